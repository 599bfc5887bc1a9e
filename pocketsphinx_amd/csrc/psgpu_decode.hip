// psgpu_decode.hip -- the first pass of a batch of utterances as ONE device pipeline:
// 16-bit PCM -> MFCC (psgpu_fe) -> 1s_c_d_dd features -> PTM senone scores (un-normalised rows) -> phone-loop search
// -> lexicon-tree search -> back-pointer tables -> best exit + backtrace, nothing through the host in between.
//
// This is the device side of what ps_decode_raw() does per utterance with -fwdflat no -bestpath no
// (pocketsphinx.c:1030-1070: ps_start_utt, ps_process_raw(full_utt), ps_end_utt, then ps_get_hyp's
// ngram_search_bp_hyp), for n_utt utterances at once.  The object owns only the buffers between the stages; the stages
// are the handles it is given (front end, scorer model, HMM context, tree search), so the reference-side binding
// (integration/psgpu_device_decode.c: tables read out of a live decoder) and a table-driven caller (bench.py: tables from
// a dump) run the same code.
#include "psgpu_internal.h"
#include <algorithm>
#include <cstring>
#include <time.h>
#include <thread>
#include <vector>

// ---- what the reference's acmod does with a live decoder's audio, as counters ------------------------------------------------------
// ps_process_raw(full_utt = FALSE) (pocketsphinx.c:1210-1246) hands its samples to acmod_process_raw in a loop; that runs the front
// end into a circular buffer of 2 x window + 1 cepstra (fe_process_frames with its overflow buffer, fe_interface.c:352-512), turns the
// buffer's content into feature frames piece by piece (acmod_process_mfcbuf / acmod_process_cep, acmod.c:565-762, each piece one
// feat_s2mfc2feat_live call -- one cmn_live call: the running mean moves only between pieces) and the searches consume every feature
// frame before the next round (ps_search_forward).  Which frames the front end makes in a call, how the cepstra fall into pieces, and
// how many feature frames each piece releases is integer bookkeeping on the sample counts alone: LiveSim walks the reference's
// counters, statement by statement, and lists the pieces; the kernels (psgpu_fe_stream_step_dev, feat_live_kernel) do the arithmetic.
// Two of the reference's oddities come with the counters and are reproduced: a decoder without a growing feature buffer (-fwdflat
// no: acmod_set_grow, ngram_search.c:147) that meets the buffer's end at an utterance's end drops the utterance's last cepstra
// (acmod.c:718-723, "FIXME"), and an utterance whose first call brings less than a frame of audio starts without the replication of
// its first frame (feat.c:1360) -- its first feature frames see the previous utterance's last cepstra in the window.
// (oracle/ref_dump.c `livefeat` dumps what the reference hands its searches for a list of chunk sizes; tests/test_streams_pcm_gpu.py.)
struct LiveSim {
    int fs = 410, sh = 160, win = 3, MA = 7, FA = 12, FA0 = 12;
    bool grow = false;
    int ov = 0;                                           // fe_t.num_overflow_samps
    int state = 0, n_mfc = 0, outidx = 0, n_feat = 0, feat_outidx = 0, nbuf = 0;      // acmod_t / feat_t counters (state: 1 started, 2 processing, 3 ended)
    std::vector<int32_t> *ops = nullptr;                  // (n, flags) pairs: flags 1 begin, 2 end, 4 statistics only
    int made = 0, feats = 0;
    bool live_seen = false;                               // a piece has gone through feat_cmn: the decoder's cmn type is CMN_LIVE from then on (feat.c:922-924)
    void setup(int frame_size, int frame_shift, int window, int pl_window, bool grow_feat)
    {
        fs = frame_size; sh = frame_shift; win = window; MA = 2 * win + 1; FA0 = FA = MA + pl_window; grow = grow_feat;
        if (grow && FA < 128) FA0 = FA = 128;
    }
    void start() { ov = 0; state = 1; n_mfc = 0; outidx = 0; n_feat = 0; feat_outidx = 0; }      // fe_start_utt + acmod_start_utt (acmod.c:407-421)
    int fe_run(long long &n_samps, int room)              // fe_process_frames_int16 (fe_interface.c:352-512): frames made
    {
        if (n_samps + ov < fs) { if (n_samps > 0) { ov += (int)n_samps; n_samps = 0; } return 0; }
        if (room < 1) return 0;
        long long consumed = 0;
        long long fc = 1 + (n_samps + ov - fs) / sh;
        if (fc > room) fc = room;
        if (ov) { const int off = fs - ov; consumed += off; n_samps -= off; ov -= sh; }
        else { consumed += fs; n_samps -= fs; }
        for (long long i = 1; i < fc; ++i) { consumed += sh; n_samps -= sh; if (ov > 0) ov -= sh; }
        if (ov <= 0) {
            long long nov = n_samps < sh ? n_samps : sh;
            ov = fs - sh;
            if (ov > consumed) ov = (int)consumed;
            ov += (int)nov;
            if (ov > 0) { consumed += nov; n_samps -= nov; }
        }
        else {
            long long nov = consumed + n_samps;
            if (nov > fs - ov) nov = fs - ov;
            ov += (int)nov;
            if (nov > consumed) { nov -= consumed; consumed += nov; n_samps -= nov; }
        }
        return (int)fc;
    }
    int live(int ncep, bool begin, bool end)              // feat_s2mfc2feat_live's counters (feat.c:1310-1420; never near LIVEBUFBLOCKSIZE here)
    {
        if (begin) nbuf = 0;
        int nb = nbuf + ncep + (end ? win : 0);
        live_seen = true;
        ops->push_back(ncep); ops->push_back((begin ? 1 : 0) | (end ? 2 : 0));
        const int nfeat = nb - win;
        if (nfeat <= 0) { nbuf = nb; return 0; }
        nbuf = nb - nfeat;
        feats += nfeat;
        return nfeat;
    }
    int process_cep(int n_frames)                         // acmod_process_cep (acmod.c:671-762)
    {
        const int orig = n_frames;
        int ncep = n_frames, nfeat = n_frames;
        if (state == 3) nfeat += win; else if (state == 1) nfeat -= win;
        if (nfeat > FA - n_feat) {
            if (grow || state == 3) FA = FA + nfeat;
            else ncep -= (nfeat - (FA - n_feat));
        }
        int inptr;
        if (grow) { inptr = feat_outidx + n_feat; while (inptr + nfeat >= FA) FA *= 2; }
        else inptr = (feat_outidx + n_feat) % FA;
        if (inptr + nfeat > FA && state == 3) return 0;   // "we can't split the last frame drop properly": the cepstra stay behind
        if (inptr + nfeat > FA) {
            const int ncep1 = FA - inptr;
            const int nf = live(ncep1, state == 1, false);
            n_feat += nf; inptr = (inptr + nf) % FA;
            n_frames -= ncep1; ncep -= ncep1;
        }
        n_feat += live(ncep, state == 1, state == 3);
        n_frames -= ncep;
        if (state == 1) state = 2;
        return orig - n_frames;
    }
    void process_mfcbuf()                                 // acmod_process_mfcbuf (acmod.c:565-599)
    {
        int ncep = n_mfc;
        if (outidx + ncep > MA) {
            int ncep1 = MA - outidx;
            const int saved = state;
            if (state == 3) state = 2;
            ncep1 = process_cep(ncep1);
            ncep -= ncep1; n_mfc -= ncep1; outidx = (outidx + ncep1) % MA;
            state = saved;
        }
        ncep = process_cep(ncep);
        n_mfc -= ncep; outidx = (outidx + ncep) % MA;
    }
    // feat_update_stats (feat.c:1422-1431): cmn_live_update -- if the decoder's cmn type has become CMN_LIVE, i.e. once a piece has gone
    // through feat_cmn outside a whole-utterance call; a decoder that has seen no live audio yet leaves its mean alone
    void stats_only() { if (live_seen) { ops->push_back(0); ops->push_back(4); } }
    void drain() { feat_outidx += n_feat; if (!grow) feat_outidx %= FA; n_feat = 0; }      // ps_search_forward: every feature frame consumed
    void process_raw(long long n_samps)                   // one ps_process_raw call (acmod_process_raw, acmod.c:601-668, in its loop)
    {
        while (n_samps) {
            int room = MA - n_mfc, inptr = (outidx + n_mfc) % MA;
            bool done = false;
            while (inptr + room > MA) {
                const int f = fe_run(n_samps, MA - inptr);
                made += f; n_mfc += f; room -= f; inptr = (inptr + f) % MA;
                if (f == 0) { done = true; break; }
            }
            if (!done) { const int f = fe_run(n_samps, room); made += f; n_mfc += f; }
            process_mfcbuf();
            drain();
        }
    }
    int end()                                             // acmod_end_utt (acmod.c:423-467); returns the tail frame's samples (0: none)
    {
        state = 3;
        int tail = 0;
        if (n_mfc < MA) {
            if (ov > 0) { tail = ov; ++made; ++n_mfc; process_mfcbuf(); }
            else stats_only();
            ov = 0;
        }
        else stats_only();
        drain();
        return tail;
    }
};

struct psgpu_decode_s;
static int dec_pcm_stream_reset(psgpu_decode_s *d, int u, bool new_decoder, hipStream_t st);

struct psgpu_decode_s {
    psgpu_decode_config_t cfg;
    uint16_t *d_ssid = nullptr, *d_ci = nullptr;
    int16_t *d_tmatid = nullptr;
    int32_t n_sen = 0, n_chain = 0, topn = 0, cepsize = 0, n_ci = 0, n1 = 0, n_emit = 0, max_words = 0;
    int32_t kind = PSGPU_SCORER_PTM, veclen = 0, raw_flag = 1;      // the scorer (psgpu_decode_config_t.scorer_kind); what the search is told about its rows
    int32_t *d_ms_id = nullptr; float *d_ms_dist = nullptr;         // the ms scorer's top-N lists (shapes that need them)
    // work buffers, grown on demand
    size_t cap_samples = 0, cap_frames = 0, cap_utt = 0, cap_bp = 0, cap_bss = 0, cap_mf = 0;
    int16_t *d_pcm = nullptr;
    float *d_cep = nullptr, *d_feat = nullptr;
    int32_t *d_off = nullptr, *d_tsc = nullptr, *d_best = nullptr, *d_pen = nullptr;
    uint8_t *d_tcw = nullptr;
    int16_t *d_rows = nullptr;
    int32_t *d_bp = nullptr, *d_bss = nullptr, *d_idx = nullptr, *d_step = nullptr, *d_res = nullptr, *d_hyp = nullptr, *d_hn = nullptr,
            *d_w1 = nullptr;
    // scores on demand: the phone loop and the search evaluate the senones they list from the scorer's top-N lists
    // (psgpu_phone_loop_run_lists_dev, psgpu_fwdtree_search_lists_dev); the senone kernel and its rows are left out
    psgpu_ptm_view_t view;
    bool lists = false, want_lists = false;
    bool compall = false;                // psgpu_decode_compallsen: every senone scored and normalised over all of them (-compallsen yes)
    // a decoder session (psgpu_decode_session): what utterance k + 1 of ONE reference decoder inherits from utterance k --
    // the scorer's last top-N lists (the seeds of the next first frame, ptm_mgau.c) and the per-state ssids of the permanent
    // multiplexed channels (hmm_clear keeps them)
    bool session = false, sess_started = false, seed_valid = false, fe_fresh = true;
    uint8_t *d_seed = nullptr;
    uint8_t *d_seed_tmp = nullptr;       // the semi-continuous scorer's slot carry-out of a call (copied to d_seed when the call wrote it)
    int32_t *d_mpx = nullptr;
    double *d_noise = nullptr;           // the front end's noise tracker (noise_stats_t: kept until ps_start_stream, not reset per utterance)
    int32_t *d_undef = nullptr;
    // table capacities per utterance = per-frame allowance x frames of the longest utterance + a constant
    // (psgpu_decode_table_capacity); grown on demand when the search reports a full table (psgpu_decode_fetch_hyps)
    int32_t bp_pf = 16, bss_pf = 320, auto_grow = 1, n_grown = 0;
    int32_t *d_mpx_in = nullptr;          // session: the state the latest search STARTED from (a repeated search needs it again)
    // psgpu_decode_second_pass: the flat-lexicon pass's tables (the first pass's capacities), its seeds, whether the latest call ran it
    int32_t *d_bp2 = nullptr, *d_bss2 = nullptr, *d_idx2 = nullptr, *d_step2 = nullptr, *d_res2 = nullptr, *d_seed2 = nullptr;
    size_t cap2_utt = 0, cap2_bp = 0, cap2_bss = 0, cap2_mf = 0;
    int32_t bp_cap2 = 0, bss_cap2 = 0;
    bool pass2 = false;
    bool last_chained = false, last_sess = false, searched = false;
    bool first_called = false;          // a psgpu_decode_first_pass* call has been made (searched: ... and a search kernel ran in it)
    int32_t lag_next = 0, last_lag = 0;   // psgpu_decode_search_lag: for the next call / what the latest call's search was given
    // the last call
    int32_t n_utt = 0, total = 0, max_frames = 0, bp_cap = 0, bss_cap = 0;
    std::vector<int32_t> frame_off;
    std::vector<int64_t> soff;
    int16_t *stage = nullptr; size_t cap_stage = 0;      // pinned: the batch's audio back to back on its way to the device (grow-only)
    // optional per-stage timing: events on the launch stream around front end | features | scorer | phone loop | search | backtrace
    bool timing = false;
    hipEvent_t ev[7] = {};
    bool ev_valid = false;
    // psgpu_decode_search_after: this object's search waits for the search of prev's latest call
    psgpu_decode_s *prev = nullptr;
    hipEvent_t ev_pre = nullptr, ev_srch = nullptr, ev_go = nullptr;
    bool srch_recorded = false, go_recorded = false;
    // psgpu_decode_front_end_ahead: the NEXT call's front end + features, run on a stream of this object's own while its latest
    // call's search is still resident -- beside the other object's scorer, whose top-N kernel uses no LDS (the spectrum kernel,
    // 4 KB of LDS a wave, gets 8 waves on a compute unit that holds two searches: run at a call's start it is 19 ms of the
    // stages' critical path against 6.4 alone)
    hipStream_t fe_stream = nullptr;
    hipEvent_t ev_wait = nullptr;                        // a blocking event: the host thread SLEEPS on it where a batch call's results are fetched (dec_wait)
    hipEvent_t ev_fe = nullptr;                          // front end ahead done
    bool fe_ahead = false;                               // ev_fe pending for the next call
    std::vector<int64_t> fe_soff;                        // the sample offsets the front end ahead was run for
    std::vector<int64_t> last_soff;                      // the latest psgpu_decode_first_pass_dev call's sample offsets
    const int16_t *fe_pcm = nullptr;
    // psgpu_decode_live_begin / _step: ONE utterance in progress, every stage going on where the previous step's frames ended --
    // the scorer's lists seeded from the last frame's (d_lseed, taking turns), the phone loop's state (d_pl_carry), the search's
    // (psgpu_fwdtree_search_resume); rows, penalties and tables of the utterance so far stay in the object's buffers
    bool live = false, live_ok = false, live_chained = false, live_mpx_copied = false;
    int32_t live_cap = 0, live_T = 0, live_S = 0, live_mode_next = 0;
    int64_t live_searched = 0;
    uint8_t *d_lseed[2] = { nullptr, nullptr };
    uint8_t *d_seed0 = nullptr;          // the session's seed as the live utterance found it (psgpu_decode_live_restart puts it back)
    bool seed0_valid = false;
    int32_t lseed_cur = 0;
    int32_t *d_pl_carry = nullptr, *d_off1 = nullptr;
    // psgpu_decode_streams_*: MANY utterances in progress, each growing at its own pace (a batch of live decoders).  Per stream the
    // frames fed (ls_T), searched (ls_S); the score rows and penalties of the frames not yet searched live in a WINDOW buffer (two,
    // taking turns: a step copies what the search has not reached yet and appends the step's new rows), ls_wbase / ls_woff = the
    // stream's first frame in it and its row there
    bool streams = false, ls_first = true;
    int32_t ls_n = 0, ls_cap = 0, ls_step = 0, ls_lag = 0, ls_cur = 0, ls_wcur = 0;
    int64_t ls_searched = 0;
    std::vector<int32_t> ls_T, ls_S, ls_wbase, ls_woff, ls_h;
    std::vector<uint8_t> ls_fresh;
    int16_t *d_win[2] = { nullptr, nullptr };
    int32_t *d_wpen[2] = { nullptr, nullptr };
    size_t win_rows = 0;
    int32_t *d_ls = nullptr;             // the step's small tables: chunk offsets [n + 1], row starts [n + 1], {scored, search to} [n][2], window map [n][5]
    uint8_t *d_sseed[2] = { nullptr, nullptr };
    int32_t *d_splc = nullptr;
    // what a stream's NEXT utterance inherits (psgpu_decode_streams_next_utt): ring slot n_hist - 1 of its scorer as the frames so far
    // left it (d_sslot, per stream; ls_slot: written since the stream's decoder was new), the multiplexed channels' ssids its latest
    // search ended with (d_smpx_out) -> what its next utterance starts from (d_smpx_in; ls_mpx: the stream's utterance takes them)
    uint8_t *d_sslot = nullptr;
    int32_t *d_smpx_in = nullptr, *d_smpx_out = nullptr;
    std::vector<uint8_t> ls_slot, ls_mpx;
    // psgpu_decode_streams_pcm_begin / _step_pcm: the streams fed with AUDIO -- per stream the front end's unframed samples with the
    // pre-emphasis prior in front (d_pcarry [n][pc_slots]), its noise tracker (d_pnoise, d_pundef), the live cepstral mean and the feature
    // window (d_pfeat: psgpu_feat_live_state_words each), and on the host the reference's buffer counters (LiveSim)
    const psgpu_feat_t *feat = nullptr;  // psgpu_decode_set_feat: the feature type computed from the front end's cepstra (NULL: 1s_c_d_dd, batch CMN)
    bool pcm_streams = false;
    int32_t pc_slots = 0, pc_fs = 0, pc_sh = 0;
    std::vector<LiveSim> pc_sim;
    std::vector<int32_t> pc_ncarry, pc_ops, pc_h;
    std::vector<float> pc_init;                          // a new decoder's feature state (cmninit)
    int16_t *d_pcarry = nullptr, *d_pwork = nullptr, *d_ppcm = nullptr;
    double *d_pnoise = nullptr;
    int32_t *d_pundef = nullptr, *d_pdesc = nullptr, *d_pops = nullptr, *d_pfoff = nullptr;
    float *d_pfeat = nullptr, *d_pcep = nullptr;
    size_t pc_work_cap = 0, pc_pcm_cap = 0, pc_ops_cap = 0, pc_cep_cap = 0;
};

static void dec_mark(psgpu_decode_s *d, int i, hipStream_t st) { if (d->timing) hipEventRecord(d->ev[i], st); }

#define DFREE(p) do { if (p) { hipFree(p); (p) = nullptr; } } while (0)

static int dec_alloc(void **p, size_t bytes)
{
    *p = nullptr;
    PSGPU_HIP(hipMalloc(p, bytes ? bytes : 4));
    return PSGPU_OK;
}

// Where the senone scores come from: full rows from the senone kernel (the default), or the phone loop and the search
// evaluating the senones they list themselves (psgpu_decode_score_mode / PSGPU_DECODE_LISTS=1).  Measured on the 512 x 30 s
// batch (profiles/, r02): without rows the scorer stage drops from 42 to 26 ms and 15.7 GB of row traffic disappear, but the
// search -- a latency-bound recurrence -- pays 12.6 k cycles per frame for list building and two dependent trips to the
// weight tables, 90 -> 116 ms, and the phone loop's preparation 0.8 -> 5.9 ms: slower in total, hence not the default.
static bool dec_can_lists(psgpu_decode_s *d)
{
    return d->kind == PSGPU_SCORER_PTM && psgpu_ptm_model_view(d->cfg.model, &d->view) == PSGPU_OK && psgpu_fwdtree_can_score_lists(d->cfg.ft, &d->view)
           && d->cfg.n_ci_list <= 256;
}
static void dec_pick_mode(psgpu_decode_s *d)
{
    static const int env_lists = [] { const char *e = getenv("PSGPU_DECODE_LISTS"); return e ? atoi(e) : 0; }();
    d->lists = (d->want_lists || env_lists) && !d->compall && !d->live && dec_can_lists(d);      // (a live utterance keeps score rows)
}

// The codeword lists the second pass's frame 0 starts from, per utterance: slot n_fast_hist - 1 of the scorer's history ring as the
// first pass left it (ptm_mgau.c:425-441) = the lists of the utterance's last frame t with t % H == H - 1, H = n_fast_hist; an
// utterance shorter than H frames never wrote that slot: a new scorer's lists, codeword = rank (:790-793).
__global__ void dec_pass2_seed_kernel(const uint8_t *__restrict__ tcw, const int32_t *__restrict__ off, int32_t total, int32_t n_chain,
                                      int32_t topn, int32_t H, int32_t *__restrict__ seed)
{
    const int u = blockIdx.x, t0 = off[u], T = off[u + 1] - t0;
    int ts = T - 1;
    while (ts >= 0 && ts % H != H - 1) --ts;
    for (int i = threadIdx.x; i < n_chain * topn; i += blockDim.x) {
        const int ch = i / topn, k = i - ch * topn;
        seed[(size_t)u * n_chain * topn + i] = ts >= 0 ? (int32_t)tcw[((size_t)ch * total + t0 + ts) * topn + k] : k;
    }
}

// psgpu_decode_streams_step: the new window of every stream = the rows (and penalties) of its old window the search has not reached
// yet + the step's new rows.  map [n][5] = {first row to keep in the old window, rows kept, first row in the step's buffer, new rows,
// first row in the new window}; one workgroup per (stream, row); dwords (n_sen is even in this mode).
__global__ __launch_bounds__(256)
void dec_window_kernel(const int32_t *__restrict__ map, int32_t n_streams, int32_t max_rows, int32_t row_dw, int32_t n_ci,
                       const uint32_t *__restrict__ win_old, const uint32_t *__restrict__ chunk, uint32_t *__restrict__ win_new,
                       const int32_t *__restrict__ pen_old, const int32_t *__restrict__ pen_chunk, int32_t *__restrict__ pen_new)
{
    const int u = blockIdx.x / max_rows, r = blockIdx.x % max_rows;
    const int32_t *const m = map + 5 * u;
    const int keep = m[1], nn = m[3];
    if (r >= keep + nn) return;
    const size_t src = r < keep ? (size_t)(m[0] + r) : (size_t)(m[2] + r - keep), dst = (size_t)(m[4] + r);
    const uint32_t *const s = (r < keep ? win_old : chunk) + src * row_dw;
    uint32_t *const o = win_new + dst * row_dw;
    for (int i = threadIdx.x; i < row_dw; i += 256) o[i] = s[i];
    const int32_t *const ps = (r < keep ? pen_old : pen_chunk) + src * n_ci;
    if ((int)threadIdx.x < n_ci) pen_new[dst * n_ci + threadIdx.x] = ps[threadIdx.x];
}

// ... ring slot n_hist - 1 of every stream's (PTM) scorer: the lists of the step's last frame ts with ts % H == H - 1, if it has one
// (tcw: the step's lists, chain-major [n_chain][total][topn]; base: the stream's frames before the step)
__global__ void dec_slot_kernel(const uint8_t *__restrict__ tcw, const int32_t *__restrict__ off1, const int32_t *__restrict__ base, int32_t total,
                                int32_t n_chain, int32_t topn, int32_t H, uint8_t *__restrict__ slot)
{
    const int u = blockIdx.x, n = off1[u + 1] - off1[u], t0 = base[u];
    int ts = t0 + n - 1;
    while (ts >= t0 && ts % H != H - 1) --ts;
    if (ts < t0) return;
    const size_t at = (size_t)off1[u] + (ts - t0);
    for (int i = threadIdx.x; i < n_chain * topn; i += blockDim.x) {
        const int ch = i / topn, k = i - ch * topn;
        slot[(size_t)u * n_chain * topn + i] = tcw[((size_t)ch * total + at) * topn + k];
    }
}

// ... and the scorer's carried lists of the streams that had no frames in the step: the batch scorer writes a carry-out for utterances
// with frames only
__global__ void dec_seed_keep_kernel(const int32_t *__restrict__ off1, int32_t n_streams, int32_t per, const uint8_t *__restrict__ in,
                                     uint8_t *__restrict__ out)
{
    const int u = blockIdx.x;
    if (off1[u + 1] != off1[u]) return;
    for (int i = threadIdx.x; i < per; i += blockDim.x) out[(size_t)u * per + i] = in[(size_t)u * per + i];
}

extern "C" {

int psgpu_decode_create(psgpu_decode_t **out, const psgpu_decode_config_t *cfg)
{
    PSGPU_REQUIRE(out && cfg && cfg->ctx && cfg->ft, "psgpu_decode_create: NULL argument");
    PSGPU_REQUIRE(cfg->scorer_kind == PSGPU_SCORER_PTM ? cfg->model != nullptr
                  : ((cfg->scorer_kind == PSGPU_SCORER_SEMI || cfg->scorer_kind == PSGPU_SCORER_MS) && cfg->scorer != nullptr),
                  "psgpu_decode_create: no scorer (model for PSGPU_SCORER_PTM, scorer for PSGPU_SCORER_SEMI / _MS)");
    PSGPU_REQUIRE(cfg->pl_ssid && cfg->pl_tmatid && cfg->ci_list && cfg->n_ci_list > 0 && cfg->pl.n_phones >= 1 && cfg->pl.n_phones <= 64
                  && cfg->pl_window >= 1, "psgpu_decode_create: the pipeline needs the phone-loop look-ahead (tables, pl_window >= 1)");
    *out = nullptr;
    int rc = psgpu_check_device();
    if (rc != PSGPU_OK) return rc;
    psgpu_decode_s *d = new psgpu_decode_s();
    d->cfg = *cfg;
    d->kind = cfg->scorer_kind;
    if (d->kind == PSGPU_SCORER_PTM) {
        d->n_sen = psgpu_ptm_n_sen(cfg->model); d->n_chain = psgpu_ptm_n_chain(cfg->model); d->topn = psgpu_ptm_topn(cfg->model);
        d->veclen = psgpu_ptm_veclen(cfg->model); d->raw_flag = 1;
    }
    else if (d->kind == PSGPU_SCORER_SEMI) {
        d->n_sen = psgpu_semi_n_sen((const psgpu_semi_model_t *)cfg->scorer); d->veclen = psgpu_semi_veclen((const psgpu_semi_model_t *)cfg->scorer);
        d->raw_flag = 3;                                 // final scores: nothing is subtracted (s2_semi_mgau.c:837-883)
        // (its carried lists: one per stream -- the sizes the seed buffers follow)
        d->n_chain = psgpu_semi_n_feat((const psgpu_semi_model_t *)cfg->scorer); d->topn = psgpu_semi_topn((const psgpu_semi_model_t *)cfg->scorer);
    }
    else {
        d->n_sen = psgpu_ms_n_sen((const psgpu_ms_model_t *)cfg->scorer); d->veclen = psgpu_ms_veclen((const psgpu_ms_model_t *)cfg->scorer);
        d->raw_flag = 1;                                 // senone_eval values: score - min over the call's list, clamped (ms_mgau.c:258-277)
    }
    d->cepsize = cfg->fe ? psgpu_fe_out_dim(cfg->fe) : 0; d->n_ci = cfg->pl.n_phones;
    d->n_emit = psgpu_hmm_n_emit_state(cfg->ctx); d->n1 = psgpu_fwdtree_n_single_phone_words(cfg->ft);
    d->max_words = cfg->max_words > 0 ? cfg->max_words : 512;
    if (d->n_sen <= 0 || d->veclen <= 0) {
        psgpu_set_error("psgpu_decode_create: the scorer reports %d senones, %d-dimensional vectors", d->n_sen, d->veclen);
        delete d;
        return PSGPU_EINVAL;
    }
    const size_t np = (size_t)cfg->pl.n_phones;
    if (dec_alloc((void **)&d->d_ssid, 2 * np) || dec_alloc((void **)&d->d_tmatid, 2 * np) || dec_alloc((void **)&d->d_ci, 2 * (size_t)cfg->n_ci_list)
        || hipMemcpy(d->d_ssid, cfg->pl_ssid, 2 * np, hipMemcpyHostToDevice) != hipSuccess
        || hipMemcpy(d->d_tmatid, cfg->pl_tmatid, 2 * np, hipMemcpyHostToDevice) != hipSuccess
        || hipMemcpy(d->d_ci, cfg->ci_list, 2 * (size_t)cfg->n_ci_list, hipMemcpyHostToDevice) != hipSuccess) {
        psgpu_set_error("psgpu_decode_create: table upload failed");
        psgpu_decode_free(d);
        return PSGPU_EHIP;
    }
    d->cfg.pl_ssid = nullptr; d->cfg.pl_tmatid = nullptr; d->cfg.ci_list = nullptr;      // (host tables are not kept)
    dec_pick_mode(d);
    *out = d;
    return PSGPU_OK;
}

void psgpu_decode_free(psgpu_decode_t *d)
{
    if (!d) return;
    if (d->stage) hipHostFree(d->stage);
    DFREE(d->d_ssid); DFREE(d->d_ci); DFREE(d->d_tmatid); DFREE(d->d_pcm); DFREE(d->d_cep); DFREE(d->d_feat); DFREE(d->d_off);
    DFREE(d->d_tsc); DFREE(d->d_best); DFREE(d->d_pen); DFREE(d->d_tcw); DFREE(d->d_rows); DFREE(d->d_bp); DFREE(d->d_bss);
    DFREE(d->d_idx); DFREE(d->d_step); DFREE(d->d_res); DFREE(d->d_hyp); DFREE(d->d_hn); DFREE(d->d_w1);
    DFREE(d->d_bp2); DFREE(d->d_bss2); DFREE(d->d_idx2); DFREE(d->d_step2); DFREE(d->d_res2); DFREE(d->d_seed2);
    DFREE(d->d_seed); DFREE(d->d_seed_tmp); DFREE(d->d_mpx); DFREE(d->d_mpx_in); DFREE(d->d_noise); DFREE(d->d_undef); DFREE(d->d_ms_id); DFREE(d->d_ms_dist);
    DFREE(d->d_lseed[0]); DFREE(d->d_lseed[1]); DFREE(d->d_seed0); DFREE(d->d_pl_carry); DFREE(d->d_off1);
    DFREE(d->d_win[0]); DFREE(d->d_win[1]); DFREE(d->d_wpen[0]); DFREE(d->d_wpen[1]); DFREE(d->d_ls); DFREE(d->d_sseed[0]); DFREE(d->d_sseed[1]);
    DFREE(d->d_splc); DFREE(d->d_sslot); DFREE(d->d_smpx_in); DFREE(d->d_smpx_out);
    DFREE(d->d_pcarry); DFREE(d->d_pwork); DFREE(d->d_ppcm); DFREE(d->d_pnoise); DFREE(d->d_pundef); DFREE(d->d_pdesc); DFREE(d->d_pops);
    DFREE(d->d_pfoff); DFREE(d->d_pfeat); DFREE(d->d_pcep);
    for (int i = 0; i < 7; ++i) if (d->ev[i]) hipEventDestroy(d->ev[i]);
    if (d->ev_pre) hipEventDestroy(d->ev_pre);
    if (d->ev_srch) hipEventDestroy(d->ev_srch);
    if (d->ev_go) hipEventDestroy(d->ev_go);
    if (d->ev_fe) hipEventDestroy(d->ev_fe);
    if (d->ev_wait) hipEventDestroy(d->ev_wait);
    if (d->fe_stream) hipStreamDestroy(d->fe_stream);
    delete d;
}

int psgpu_decode_score_mode(psgpu_decode_t *d, int32_t lists)
{
    PSGPU_REQUIRE(d, "psgpu_decode_score_mode: NULL argument");
    d->want_lists = lists != 0;
    PSGPU_REQUIRE(!lists || dec_can_lists(d), "psgpu_decode_score_mode: this model pair cannot score from lists (psgpu_fwdtree_can_score_lists)");
    dec_pick_mode(d);
    return PSGPU_OK;
}

int psgpu_decode_session(psgpu_decode_t *d, int32_t on)
{
    PSGPU_REQUIRE(d, "psgpu_decode_session: NULL argument");
    d->session = on != 0;
    d->sess_started = false; d->seed_valid = false; d->fe_fresh = true;
    return PSGPU_OK;
}

static int dec_session_buffers(psgpu_decode_s *d)
{
    int rc;
    if (!d->d_seed) {
        if ((rc = dec_alloc((void **)&d->d_seed, (size_t)d->n_chain * d->topn)) || (rc = dec_alloc((void **)&d->d_seed_tmp, (size_t)d->n_chain * d->topn))
            || (rc = dec_alloc((void **)&d->d_mpx, 4 * (size_t)std::max(1, psgpu_fwdtree_n_mpx_channels(d->cfg.ft)) * d->n_emit))
            || (rc = dec_alloc((void **)&d->d_mpx_in, 4 * (size_t)std::max(1, psgpu_fwdtree_n_mpx_channels(d->cfg.ft)) * d->n_emit))
            || (rc = dec_alloc((void **)&d->d_noise, 8 * 4 * 64)) || (rc = dec_alloc((void **)&d->d_undef, 4)))
            return rc;
        d->fe_fresh = true;
    }
    return PSGPU_OK;
}

int psgpu_decode_session_set(psgpu_decode_t *d, const uint8_t *seed_cw, const int32_t *mpx_ssid, void *stream)
{
    PSGPU_REQUIRE(d && d->session, "psgpu_decode_session_set: not in session mode");
    int rc = dec_session_buffers(d);
    if (rc != PSGPU_OK) return rc;
    hipStream_t st = (hipStream_t)stream;
    if (seed_cw) {
        PSGPU_HIP(hipMemcpyAsync(d->d_seed, seed_cw, (size_t)d->n_chain * d->topn, hipMemcpyHostToDevice, st));
        d->seed_valid = true;
    }
    else d->seed_valid = false;
    if (mpx_ssid) {
        PSGPU_HIP(hipMemcpyAsync(d->d_mpx, mpx_ssid, 4 * (size_t)psgpu_fwdtree_n_mpx_channels(d->cfg.ft) * d->n_emit, hipMemcpyHostToDevice, st));
        d->sess_started = true;
    }
    else d->sess_started = false;
    PSGPU_HIP(hipStreamSynchronize(st));
    return PSGPU_OK;
}

int psgpu_decode_session_get(psgpu_decode_t *d, uint8_t *seed_cw, int32_t *seed_valid, int32_t *mpx_ssid, void *stream)
{
    PSGPU_REQUIRE(d && d->session && d->sess_started, "psgpu_decode_session_get: no session utterance decoded yet");
    hipStream_t st = (hipStream_t)stream;
    if (seed_valid) *seed_valid = d->seed_valid ? 1 : 0;
    if (seed_cw && d->seed_valid) PSGPU_HIP(hipMemcpyAsync(seed_cw, d->d_seed, (size_t)d->n_chain * d->topn, hipMemcpyDeviceToHost, st));
    if (mpx_ssid)
        PSGPU_HIP(hipMemcpyAsync(mpx_ssid, d->d_mpx, 4 * (size_t)psgpu_fwdtree_n_mpx_channels(d->cfg.ft) * d->n_emit, hipMemcpyDeviceToHost, st));
    PSGPU_HIP(hipStreamSynchronize(st));
    return PSGPU_OK;
}

int psgpu_decode_search_after(psgpu_decode_t *d, psgpu_decode_t *prev)
{
    PSGPU_REQUIRE(d != nullptr && d != prev, "psgpu_decode_search_after: bad argument");
    for (psgpu_decode_s *q : { d, prev })
        if (q && !q->ev_pre) {
            PSGPU_HIP(hipEventCreateWithFlags(&q->ev_pre, hipEventDisableTiming));
            PSGPU_HIP(hipEventCreateWithFlags(&q->ev_srch, hipEventDisableTiming));
            PSGPU_HIP(hipEventCreateWithFlags(&q->ev_go, hipEventDisableTiming));
            PSGPU_HIP(hipEventCreateWithFlags(&q->ev_fe, hipEventDisableTiming));
        }
    d->prev = prev;
    return PSGPU_OK;
}

int psgpu_decode_wait_scored(psgpu_decode_t *d)
{
    PSGPU_REQUIRE(d && d->ev_pre, "psgpu_decode_wait_scored: psgpu_decode_search_after first");
    PSGPU_HIP(hipEventSynchronize(d->ev_pre));
    return PSGPU_OK;
}

// -compallsen yes: acmod_score has the scorer evaluate EVERY senone and normalise over all of them (acmod.c:1098-1128, the scorers'
// compallsen branches: ptm_mgau.c:393-400 over the whole array, ms_mgau.c:213-236), whatever the searches list; the phone loop and the
// search then take the rows as final scores (the mode the semi-continuous scorer always has).
int psgpu_decode_compallsen(psgpu_decode_t *d, int32_t on)
{
    PSGPU_REQUIRE(d && d->kind != PSGPU_SCORER_SEMI, "psgpu_decode_compallsen: NULL argument, or the semi-continuous scorer (its scores are final anyway)");
    PSGPU_REQUIRE(!(on && d->want_lists), "psgpu_decode_compallsen: scoring from lists (psgpu_decode_score_mode) evaluates the listed senones only");
    d->compall = on != 0;
    d->raw_flag = on ? 3 : 1;
    dec_pick_mode(d);
    return PSGPU_OK;
}

int psgpu_decode_set_scorer(psgpu_decode_t *d, void *scorer)
{
    PSGPU_REQUIRE(d && scorer, "psgpu_decode_set_scorer: NULL argument");
    PSGPU_REQUIRE(d->kind != PSGPU_SCORER_PTM, "psgpu_decode_set_scorer: the pipeline holds a PTM scorer (psgpu_decode_set_model)");
    const int32_t ns = d->kind == PSGPU_SCORER_SEMI ? psgpu_semi_n_sen((const psgpu_semi_model_t *)scorer) : psgpu_ms_n_sen((const psgpu_ms_model_t *)scorer);
    const int32_t vl = d->kind == PSGPU_SCORER_SEMI ? psgpu_semi_veclen((const psgpu_semi_model_t *)scorer) : psgpu_ms_veclen((const psgpu_ms_model_t *)scorer);
    PSGPU_REQUIRE(ns == d->n_sen && vl == d->veclen, "psgpu_decode_set_scorer: the model has another shape");
    if (scorer != d->cfg.scorer && d->kind == PSGPU_SCORER_MS) {          // (list buffers follow the model's top-N)
        DFREE(d->d_ms_id); DFREE(d->d_ms_dist); d->cap_frames = 0;
    }
    d->cfg.scorer = scorer;
    return PSGPU_OK;
}

int psgpu_decode_set_model(psgpu_decode_t *d, psgpu_ptm_model_t *model)
{
    PSGPU_REQUIRE(d && model, "psgpu_decode_set_model: NULL argument");
    PSGPU_REQUIRE(d->kind == PSGPU_SCORER_PTM, "psgpu_decode_set_model: the pipeline was created with another scorer");
    PSGPU_REQUIRE(psgpu_ptm_n_sen(model) == d->n_sen && psgpu_ptm_n_chain(model) == d->n_chain && psgpu_ptm_topn(model) == d->topn,
                  "psgpu_decode_set_model: the model has another shape");
    d->cfg.model = model;
    dec_pick_mode(d);
    return PSGPU_OK;
}

// buffers for n_utt utterances of `total` frames in all, the longest `mf` frames
static int dec_grow(psgpu_decode_s *d, size_t n_utt, size_t total, size_t mf, hipStream_t st)
{
    // table capacities follow the longest utterance: the bundled recordings write 4-6 back-pointers and 30-80 score-stack
    // entries per frame with a 100-word vocabulary (tests/golden/fwdtree_trace_*), 30 and 800 with 134,865 words (the
    // benchmark's synthetic utterances); a full table ends the utterance with status 1, and psgpu_decode_fetch_hyps then
    // repeats the search with larger tables (the reference doubles its tables on demand, ngram_search.c:449-463)
    const size_t bp_cap = (size_t)d->bp_pf * mf + 2048, bss_cap = (size_t)d->bss_pf * mf + 8192;
    bool waited = false;
    auto wait = [&]() { if (!waited) { hipStreamSynchronize(st); waited = true; } };
    if (total > d->cap_frames || (!d->lists && !d->d_rows)) {         // (.. or the model changed to one that needs score rows)
        wait();
        const size_t t = std::max(total + total / 8 + 64, d->cap_frames), ne = t * d->n_chain * d->topn;
        DFREE(d->d_cep); DFREE(d->d_feat); DFREE(d->d_tsc); DFREE(d->d_tcw); DFREE(d->d_rows); DFREE(d->d_best); DFREE(d->d_pen);
        DFREE(d->d_ms_id); DFREE(d->d_ms_dist);
        d->cap_frames = 0;
        int rc;
        const size_t ms_ent = (d->kind == PSGPU_SCORER_MS && psgpu_ms_batch_needs_lists((const psgpu_ms_model_t *)d->cfg.scorer))
                              ? t * (size_t)psgpu_ms_list_entries_per_frame((const psgpu_ms_model_t *)d->cfg.scorer) : 0;
        if ((rc = dec_alloc((void **)&d->d_cep, 4 * t * d->cepsize)) || (rc = dec_alloc((void **)&d->d_feat, 4 * t * d->veclen))
            || (rc = dec_alloc((void **)&d->d_tsc, 4 * ne)) || (rc = dec_alloc((void **)&d->d_tcw, ne))
            || (!d->lists && (rc = dec_alloc((void **)&d->d_rows, 2 * t * d->n_sen + 64))) || (!d->lists && (rc = dec_alloc((void **)&d->d_best, 4 * t)))
            || (rc = dec_alloc((void **)&d->d_pen, 4 * t * d->n_ci))
            || (ms_ent && ((rc = dec_alloc((void **)&d->d_ms_id, 4 * ms_ent)) || (rc = dec_alloc((void **)&d->d_ms_dist, 4 * ms_ent)))))
            return rc;
        d->cap_frames = t;
    }
    if (n_utt > d->cap_utt || bp_cap > d->cap_bp || bss_cap > d->cap_bss || mf > d->cap_mf) {
        wait();
        const size_t nu = std::max(n_utt, d->cap_utt), cb = std::max(bp_cap, d->cap_bp), cs = std::max(bss_cap, d->cap_bss),
                     cm = std::max(mf, d->cap_mf);
        if (d->cfg.fe) psgpu_fe_offsets_dirty(d->cfg.fe);
        DFREE(d->d_off); DFREE(d->d_bp); DFREE(d->d_bss); DFREE(d->d_idx); DFREE(d->d_step); DFREE(d->d_res); DFREE(d->d_hyp); DFREE(d->d_hn);
        DFREE(d->d_w1);
        d->cap_utt = 0;
        int rc;
        if ((rc = dec_alloc((void **)&d->d_off, 4 * (nu + 1))) || (rc = dec_alloc((void **)&d->d_bp, 4 * nu * 10 * cb))
            || (rc = dec_alloc((void **)&d->d_bss, 4 * nu * cs)) || (rc = dec_alloc((void **)&d->d_idx, 4 * nu * (cm + 2)))
            || (rc = dec_alloc((void **)&d->d_step, 4 * nu * (cm ? cm : 1) * 4)) || (rc = dec_alloc((void **)&d->d_res, 4 * nu * 8))
            || (rc = dec_alloc((void **)&d->d_hyp, 4 * nu * d->max_words * 4)) || (rc = dec_alloc((void **)&d->d_hn, 4 * nu * 4))
            || (rc = dec_alloc((void **)&d->d_w1, 4 * nu * (size_t)std::max(1, d->n1) * d->n_emit)))
            return rc;
        d->cap_utt = nu; d->cap_bp = cb; d->cap_bss = cs; d->cap_mf = cm;
    }
    return PSGPU_OK;
}

// the tree search (+ backtrace) of the latest call's utterances on the scores / penalties in the object's buffers
static int dec_search(psgpu_decode_s *d, int32_t n_utt, size_t total, size_t mf, hipStream_t st)
{
    int rc;
    const bool chained = d->last_chained, sess = d->last_sess;
    // the hypotheses are the search kernel's last step
    if ((rc = psgpu_fwdtree_hyp_out(d->cfg.ft, d->d_hyp, d->d_hn, d->max_words))) return rc;
    if ((rc = psgpu_fwdtree_search_lag(d->cfg.ft, d->last_lag))) return rc;
    if ((rc = psgpu_fwdtree_search_resume(d->cfg.ft, d->live_mode_next))) return rc;
    d->live_mode_next = 0;
    // (the idx rows are per utterance max_frames + 2 wide: the stride of this call, not of the allocation)
    if (d->lists)
        rc = psgpu_fwdtree_search_lists_dev(d->cfg.ft, &d->view, d->d_tsc, d->d_tcw, (int32_t)total, d->d_pen, d->d_off, n_utt, (int32_t)mf,
                                            d->bp_cap, d->bss_cap, d->d_bp, d->d_bss, d->d_idx, d->d_step, d->d_res, d->cfg.pl_window, d->d_w1,
                                            chained ? d->d_mpx_in : nullptr, sess ? d->d_mpx : nullptr, st);
    else
        rc = psgpu_fwdtree_search_session_dev(d->cfg.ft, d->d_rows, d->n_sen, d->d_pen, d->d_off, n_utt, (int32_t)mf, d->bp_cap, d->bss_cap,
                                              d->d_bp, d->d_bss, d->d_idx, d->d_step, d->d_res, d->raw_flag, d->cfg.pl_window, d->d_w1,
                                              chained ? d->d_mpx_in : nullptr, sess ? d->d_mpx : nullptr, st);
    if (rc == PSGPU_OK) d->searched = true;
    return rc;
}

// scores -> phone loop -> tree search -> backtrace on the features in d_feat / the frame offsets in d_off
static int dec_from_feat(psgpu_decode_s *d, int32_t n_utt, size_t total, size_t mf, hipStream_t st)
{
    int rc;
    // session: one utterance per call, chained to the call before
    const bool sess = d->session && n_utt == 1;
    if (sess && (rc = dec_session_buffers(d))) return rc;
    const bool chained = sess && d->sess_started;
    dec_mark(d, 2, st);
    if (d->kind == PSGPU_SCORER_PTM)
        rc = psgpu_ptm_score_batch_dev(d->cfg.model, d->d_feat, d->d_off, n_utt, (int32_t)total, chained && d->seed_valid ? d->d_seed : nullptr,
                                       nullptr, d->d_tsc, d->d_tcw, d->lists ? nullptr : d->d_rows, d->lists ? nullptr : d->d_best,
                                       d->compall ? 0u : PSGPU_PTM_RAW_SCORES, st);
    else if (d->kind == PSGPU_SCORER_SEMI) {
        // every utterance from a new scorer's lists, frames numbered from 0 -- a session's next utterance from ring slot n_hist - 1 as
        // the one before left it, like the PTM scorer's (s2_semi_mgau.c:853-860; n_topn_hist = pl_window + 2, :1301)
        const int H = d->cfg.pl_window + 2;
        rc = psgpu_semi_score_batch_carry_dev((psgpu_semi_model_t *)d->cfg.scorer, d->d_feat, d->d_off, n_utt, (int32_t)total,
                                              chained && d->seed_valid ? d->d_seed : nullptr, nullptr, sess ? d->d_seed_tmp : nullptr, H, nullptr,
                                              d->d_rows, st);
        if (rc == PSGPU_OK && sess && (int)total >= H) {  // (a frame ts with ts % H == H - 1 exists: the slot was written)
            PSGPU_HIP(hipMemcpyAsync(d->d_seed, d->d_seed_tmp, (size_t)d->n_chain * d->topn, hipMemcpyDeviceToDevice, st));
            d->seed_valid = true;
        }
    }
    else                                                 // no time dependence: frames of all utterances back to back
        rc = d->compall ? psgpu_ms_score_batch_dev((psgpu_ms_model_t *)d->cfg.scorer, d->d_feat, (int32_t)total, d->d_ms_id, d->d_ms_dist, d->d_rows, st)
                        : psgpu_ms_score_batch_raw_dev((psgpu_ms_model_t *)d->cfg.scorer, d->d_feat, (int32_t)total, d->d_ms_id, d->d_ms_dist, d->d_rows, st);
    if (rc) return rc;
    if (sess && d->kind == PSGPU_SCORER_PTM) {
        // what seeds the next utterance's first frame: ptm_mgau_frame_eval copies frame 0's initial lists from slot
        // n_fast_hist - 1 of its history ring (ptm_mgau.c:425-441), H = n_fast_hist = pl_window + 2 (:865); that slot was last
        // written by the last frame ts with ts % H == H - 1 -- not by the last frame.  Shorter utterances leave it alone.
        const int H = d->cfg.pl_window + 2, T = (int)total;
        int ts = T - 1;
        while (ts >= 0 && ts % H != H - 1) --ts;
        if (ts >= 0) {
            PSGPU_HIP(hipMemcpy2DAsync(d->d_seed, (size_t)d->topn, d->d_tcw + (size_t)ts * d->topn, (size_t)T * d->topn, (size_t)d->topn,
                                       (size_t)d->n_chain, hipMemcpyDeviceToDevice, st));
            d->seed_valid = true;
        }
    }
    dec_mark(d, 3, st);
    if (d->lists)
        rc = psgpu_phone_loop_run_lists_dev(d->cfg.ctx, &d->cfg.pl, d->d_ssid, d->d_tmatid, d->d_ci, d->cfg.n_ci_list, &d->view, d->d_tsc,
                                            d->d_tcw, d->d_off, n_utt, (int32_t)total, d->d_pen, nullptr, nullptr, st);
    else
        rc = psgpu_phone_loop_run_dev(d->cfg.ctx, &d->cfg.pl, d->d_ssid, d->d_tmatid, d->raw_flag == 3 ? nullptr : d->d_ci,
                                      d->raw_flag == 3 ? 0 : d->cfg.n_ci_list, d->d_rows, d->n_sen,
                                      nullptr, d->d_off, n_utt, (int32_t)total, d->d_pen, nullptr, nullptr, st);
    if (rc) return rc;
    dec_mark(d, 4, st);
    if (d->ev_pre) {
        PSGPU_HIP(hipEventRecord(d->ev_pre, st));            // psgpu_decode_wait_scored
        // (PSGPU_DECODE_SEARCH_OVERLAP=1: searches of the two objects may be resident together -- a measuring knob for builds of
        //  the search kernel that leave room for a third workgroup per compute unit)
        static const int overlap = [] { const char *e = getenv("PSGPU_DECODE_SEARCH_OVERLAP"); return e ? atoi(e) : 0; }();
        if (!overlap && d->prev && d->prev->srch_recorded) PSGPU_HIP(hipStreamWaitEvent(st, d->prev->ev_srch, 0));
        // (... and for the other object's front end ahead, if one is on its way: the search is dispatched onto a device whose LDS
        //  nobody else holds)
        if (d->prev && d->prev->fe_ahead) PSGPU_HIP(hipStreamWaitEvent(st, d->prev->ev_fe, 0));
        PSGPU_HIP(hipEventRecord(d->ev_go, st));             // "this call's search is being dispatched": the other object's next call waits for it
        d->go_recorded = true;
    }
    dec_mark(d, 5, st);                                  // (4 -> 5: waiting for the other object's search, if any)
    if (sess && chained)                                 // (what this search starts from: a repeated search starts from it again)
        PSGPU_HIP(hipMemcpyAsync(d->d_mpx_in, d->d_mpx, 4 * (size_t)psgpu_fwdtree_n_mpx_channels(d->cfg.ft) * d->n_emit,
                                 hipMemcpyDeviceToDevice, st));
    d->last_chained = chained; d->last_sess = sess;
    d->last_lag = d->lag_next; d->lag_next = 0;
    rc = dec_search(d, n_utt, total, mf, st);
    if (rc) return rc;
    if (d->ev_srch) { PSGPU_HIP(hipEventRecord(d->ev_srch, st)); d->srch_recorded = true; }
    if (sess) d->sess_started = true;
    dec_mark(d, 6, st);
    return rc;
}

// A front end run ahead (psgpu_decode_front_end_ahead) writes d_cep / d_feat / d_off on its own stream: every entry point that
// writes or re-reads those buffers first orders its stream after it and forgets it (psgpu_decode_first_pass_dev uses its result
// when the input is the one it was run for).
static int dec_settle_fe_ahead(psgpu_decode_s *d, hipStream_t st)
{
    if (d->fe_ahead) { PSGPU_HIP(hipStreamWaitEvent(st, d->ev_fe, 0)); d->fe_ahead = false; }
    return PSGPU_OK;
}

// cepstra -> the scorer's feature vectors: the type psgpu_decode_set_feat installed, else en-us's 1s_c_d_dd with batch CMN
static int dec_features(psgpu_decode_s *d, int32_t n_utt, hipStream_t st)
{
    if (d->feat) return psgpu_feat_compute_dev(d->feat, d->d_cep, d->d_off, n_utt, d->d_feat, st);
    return psgpu_feat_1s_c_d_dd_dev(d->d_cep, d->d_off, n_utt, d->cepsize, d->d_feat, st);
}

int psgpu_decode_set_feat(psgpu_decode_t *d, const psgpu_feat_t *feat)
{
    PSGPU_REQUIRE(d, "psgpu_decode_set_feat: NULL argument");
    PSGPU_REQUIRE(!feat || (d->cfg.fe && psgpu_feat_cepsize(feat) == d->cepsize && psgpu_feat_out_dim(feat) == d->veclen),
                  "psgpu_decode_set_feat: the feature type takes %d cepstra and makes %d-dimensional vectors; the front end makes %d, the scorer takes %d",
                  feat ? psgpu_feat_cepsize(feat) : 0, feat ? psgpu_feat_out_dim(feat) : 0, d->cepsize, d->veclen);
    d->feat = feat;
    return PSGPU_OK;
}

int psgpu_decode_first_pass_dev(psgpu_decode_t *d, const int16_t *pcm_dev, const int64_t *samp_off, int32_t n_utt, void *stream)
{
    PSGPU_REQUIRE(d && n_utt >= 0 && (n_utt == 0 || (pcm_dev && samp_off)), "psgpu_decode_first_pass_dev: bad argument");
    PSGPU_REQUIRE(d->cfg.fe && (d->feat || d->veclen == 3 * d->cepsize),
                  "psgpu_decode_first_pass_dev: from PCM the pipeline computes 1s_c_d_dd vectors of %d cepstra; the scorer takes %d-dimensional "
                  "vectors (other feature types: psgpu_decode_set_feat, or psgpu_decode_first_pass_feat)", d->cepsize, d->veclen);
    hipStream_t st = (hipStream_t)stream;
    d->n_utt = n_utt; d->total = 0; d->max_frames = 0; d->searched = false; d->pass2 = false; d->first_called = true;
    d->live = false; d->streams = false;
    d->frame_off.assign((size_t)n_utt + 1, 0);
    if (n_utt == 0) return PSGPU_OK;
    size_t total = 0, mf = 0;
    for (int u = 0; u < n_utt; ++u) {
        PSGPU_REQUIRE(samp_off[u + 1] >= samp_off[u], "psgpu_decode_first_pass_dev: sample offsets must not decrease");
        const size_t t = (size_t)psgpu_fe_n_frames(d->cfg.fe, samp_off[u + 1] - samp_off[u]);
        total += t; mf = std::max(mf, t);
    }
    PSGPU_REQUIRE(total < 0x7fffff00u, "psgpu_decode_first_pass_dev: %zu frames in one call", total);
    int rc = dec_grow(d, (size_t)n_utt, total ? total : 1, mf, st);
    if (rc != PSGPU_OK) return rc;
    d->total = (int32_t)total; d->max_frames = (int32_t)mf;
    d->bp_cap = (int32_t)d->cap_bp; d->bss_cap = (int32_t)d->cap_bss;
    d->ev_valid = false;
    d->last_soff.assign(samp_off, samp_off + n_utt + 1);
    {   // the frame offsets on the host (the front end's call fills them too; a front end run ahead has gone before)
        int32_t acc = 0;
        for (int u = 0; u < n_utt; ++u) { d->frame_off[u] = acc; acc += (int32_t)psgpu_fe_n_frames(d->cfg.fe, samp_off[u + 1] - samp_off[u]); }
        d->frame_off[n_utt] = acc;
    }
    // taking turns with another object (psgpu_decode_search_after): this call's first stages start when the other object's
    // search has been dispatched -- onto a device that runs nothing else at that moment, so that all its workgroups are placed
    // at once.  A search kernel dispatched while other kernels hold part of the compute units' LDS gets one workgroup per
    // compute unit instead of two and takes twice as long (profiles/r03_overlap.txt).
    if (d->prev && d->prev->go_recorded) PSGPU_HIP(hipStreamWaitEvent(st, d->prev->ev_go, 0));
    dec_mark(d, 0, st);
    // the front end of exactly this input has been run ahead (psgpu_decode_front_end_ahead): cepstra, features and offsets are
    // in the object's buffers (or on their way: the event)
    const bool ahead = d->fe_ahead && d->fe_pcm == pcm_dev && d->fe_soff.size() == (size_t)n_utt + 1
                       && memcmp(d->fe_soff.data(), samp_off, sizeof(int64_t) * ((size_t)n_utt + 1)) == 0 && !(d->session && n_utt == 1);
    if ((rc = dec_settle_fe_ahead(d, st))) return rc;                    // (used or not: nothing of it may still be running)
    if (ahead && total > 0) {
        dec_mark(d, 1, st);
        if ((rc = dec_from_feat(d, n_utt, total, mf, st))) return rc;
        d->ev_valid = d->timing;
        return PSGPU_OK;
    }
    // session: the noise tracker of the reference's front end lives until ps_start_stream (fe_start_utt, fe_interface.c:318-326,
    // does not reset it): utterance k + 1's spectra are cleaned with what utterance k left
    const bool sess = d->session && n_utt == 1;
    if (sess) {
        if ((rc = dec_session_buffers(d))) return rc;
        if (d->fe_fresh) {
            const int32_t one = 1;                           // "undefined": initialise from the first frame (fe_reset_noisestats)
            PSGPU_HIP(hipMemcpyAsync(d->d_undef, &one, 4, hipMemcpyHostToDevice, st));
            PSGPU_HIP(hipStreamSynchronize(st));
            d->fe_fresh = false;
        }
    }
    if ((rc = psgpu_fe_process_utts_dev(d->cfg.fe, pcm_dev, samp_off, n_utt, sess ? d->d_noise : nullptr, sess ? d->d_undef : nullptr,
                                        d->d_cep, d->d_off, d->frame_off.data(), st)))
        return rc;
    dec_mark(d, 1, st);
    if (total == 0) {                                   // nothing but empty utterances: empty results
        PSGPU_HIP(hipMemsetAsync(d->d_res, 0, 4 * (size_t)n_utt * 8, st));
        PSGPU_HIP(hipMemsetAsync(d->d_hn, 0, 4 * (size_t)n_utt * 4, st));
        return PSGPU_OK;
    }
    if ((rc = dec_features(d, n_utt, st))) return rc;
    if ((rc = dec_from_feat(d, n_utt, total, mf, st))) return rc;
    d->ev_valid = d->timing;
    return PSGPU_OK;
}

// feature vectors in, from the host: feat [total][3 * cepsize] as feat_s2mfc2feat_live leaves them (acmod->feat_buf),
// frame_off [n_utt + 1].  For a binding whose host has already run the reference's own front end (ps_process_raw).
int psgpu_decode_first_pass_feat(psgpu_decode_t *d, const float *feat, const int32_t *frame_off, int32_t n_utt, void *stream)
{
    PSGPU_REQUIRE(d && n_utt >= 0 && (n_utt == 0 || (feat && frame_off)), "psgpu_decode_first_pass_feat: bad argument");
    hipStream_t st = (hipStream_t)stream;
    d->n_utt = n_utt; d->total = 0; d->max_frames = 0; d->searched = false; d->pass2 = false; d->first_called = true;
    d->live = false; d->streams = false;
    d->frame_off.assign(frame_off, frame_off + (n_utt ? n_utt + 1 : 0));
    if (n_utt == 0) { d->frame_off.assign(1, 0); return PSGPU_OK; }
    PSGPU_REQUIRE(frame_off[0] == 0, "psgpu_decode_first_pass_feat: frame offsets start at 0");
    size_t mf = 0;
    for (int u = 0; u < n_utt; ++u) {
        PSGPU_REQUIRE(frame_off[u + 1] >= frame_off[u], "psgpu_decode_first_pass_feat: frame offsets must not decrease");
        mf = std::max(mf, (size_t)(frame_off[u + 1] - frame_off[u]));
    }
    const size_t total = (size_t)frame_off[n_utt];
    int rc = dec_settle_fe_ahead(d, st);
    if (rc != PSGPU_OK) return rc;
    if ((rc = dec_grow(d, (size_t)n_utt, total ? total : 1, mf, st)) != PSGPU_OK) return rc;
    d->total = (int32_t)total; d->max_frames = (int32_t)mf;
    d->bp_cap = (int32_t)d->cap_bp; d->bss_cap = (int32_t)d->cap_bss;
    PSGPU_HIP(hipStreamSynchronize(st));                 // frame_off / feat are the caller's: copied before returning
    if (d->cfg.fe) psgpu_fe_offsets_dirty(d->cfg.fe);
    PSGPU_HIP(hipMemcpyAsync(d->d_off, frame_off, 4 * ((size_t)n_utt + 1), hipMemcpyHostToDevice, st));
    if (total) PSGPU_HIP(hipMemcpyAsync(d->d_feat, feat, 4 * total * d->veclen, hipMemcpyHostToDevice, st));
    PSGPU_HIP(hipStreamSynchronize(st));
    if (total == 0) {
        PSGPU_HIP(hipMemsetAsync(d->d_res, 0, 4 * (size_t)n_utt * 8, st));
        PSGPU_HIP(hipMemsetAsync(d->d_hn, 0, 4 * (size_t)n_utt * 4, st));
        return PSGPU_OK;
    }
    return dec_from_feat(d, n_utt, total, mf, st);
}

// The next call's front end, ahead of the call: see psgpu_decode_s::fe_stream.  Runs only for an input of the latest call's shape
// (same utterance count and sample offsets: the buffers fit, and the frame offsets the resident search still reads do not change) --
// otherwise it does nothing and the call runs its own front end, as it does for a session's single utterances.
int psgpu_decode_front_end_ahead(psgpu_decode_t *d, const int16_t *pcm_dev, const int64_t *samp_off, int32_t n_utt, int32_t *started)
{
    PSGPU_REQUIRE(d && pcm_dev && samp_off && n_utt > 0, "psgpu_decode_front_end_ahead: bad argument");
    if (started) *started = 0;
    if (!d->cfg.fe || !d->prev || !d->ev_fe || d->fe_ahead || (d->session && n_utt == 1) || d->n_utt != n_utt || d->total <= 0
        || d->last_soff.size() != (size_t)n_utt + 1 || memcmp(d->last_soff.data(), samp_off, sizeof(int64_t) * ((size_t)n_utt + 1)) != 0)
        return PSGPU_OK;
    if (!d->fe_stream) {
        void *s = nullptr;
        int rc = psgpu_stream_create_dedicated(&s);
        if (rc != PSGPU_OK) return rc;
        d->fe_stream = (hipStream_t)s;
    }
    hipStream_t fs = d->fe_stream;
    // after this object's latest search has been dispatched (its scorer and phone loop, which read the features, are over; the
    // search was placed on a device whose LDS nobody else held).  The caller issues this BEFORE the other object's next call: that
    // call's search dispatch then waits for this front end (dec_from_feat), and its scorer -- whose top-N kernel uses no LDS -- is
    // what runs beside it
    if (d->go_recorded) PSGPU_HIP(hipStreamWaitEvent(fs, d->ev_go, 0));
    std::vector<int32_t> fo((size_t)n_utt + 1);
    int rc = psgpu_fe_process_utts_dev(d->cfg.fe, pcm_dev, samp_off, n_utt, nullptr, nullptr, d->d_cep, d->d_off, fo.data(), fs);
    if (rc) return rc;
    if ((rc = dec_features(d, n_utt, fs))) return rc;
    PSGPU_HIP(hipEventRecord(d->ev_fe, fs));
    d->fe_ahead = true; d->fe_pcm = pcm_dev; d->fe_soff.assign(samp_off, samp_off + n_utt + 1);
    if (started) *started = 1;
    return PSGPU_OK;
}

int psgpu_decode_first_pass(psgpu_decode_t *d, const int16_t *const pcm[], const size_t n[], int32_t n_utt, void *stream)
{
    PSGPU_REQUIRE(d && n_utt >= 0 && (n_utt == 0 || (pcm && n)), "psgpu_decode_first_pass: bad argument");
    hipStream_t st = (hipStream_t)stream;
    d->soff.assign((size_t)n_utt + 1, 0);
    for (int u = 0; u < n_utt; ++u) d->soff[u + 1] = d->soff[u] + (int64_t)n[u];
    const size_t ns = (size_t)d->soff[n_utt];
    if (ns > d->cap_samples) {
        PSGPU_HIP(hipStreamSynchronize(st));
        DFREE(d->d_pcm); d->cap_samples = 0;
        int rc = dec_alloc((void **)&d->d_pcm, 2 * (ns + ns / 8 + 64));
        if (rc != PSGPU_OK) return rc;
        d->cap_samples = ns + ns / 8 + 64;
    }
    PSGPU_HIP(hipStreamSynchronize(st));                 // the staging buffer of the previous call may still be in flight
    if (ns > d->cap_stage) {                             // (pinned staging: through pageable memory the copy of 512 x 30 s -- 491 MB -- took longer than the search)
        if (d->stage) hipHostFree(d->stage);
        d->stage = nullptr; d->cap_stage = 0;
        PSGPU_HIP(hipHostMalloc((void **)&d->stage, 2 * (ns + ns / 8 + 64), hipHostMallocDefault));
        d->cap_stage = ns + ns / 8 + 64;
    }
    {   // the utterances side by side into the staging buffer: a few host threads (one memcpy stream moves ~10 GB/s)
        const int n_thr = (int)std::max(1u, std::min({ std::thread::hardware_concurrency(), 8u, (unsigned)(ns >> 23) }));
        auto work = [&](int t) { for (int u = t; u < n_utt; u += n_thr) if (n[u]) memcpy(d->stage + d->soff[u], pcm[u], 2 * n[u]); };
        if (n_thr <= 1) work(0);
        else {
            std::vector<std::thread> thr;
            for (int t = 1; t < n_thr; ++t) thr.emplace_back(work, t);
            work(0);
            for (auto &t : thr) t.join();
        }
    }
    if (ns) PSGPU_HIP(hipMemcpyAsync(d->d_pcm, d->stage, 2 * ns, hipMemcpyHostToDevice, st));
    return psgpu_decode_first_pass_dev(d, d->d_pcm, d->soff.data(), n_utt, st);
}

// ---- an utterance in progress ----------------------------------------------------------------------------------------------------
// the search's resume mode of a live utterance's step
static int dec_live_mode(psgpu_decode_s *d, bool resume)
{
    (void)d;
    return PSGPU_SEARCH_KEEP | (resume ? PSGPU_SEARCH_RESUME : 0);
}

static int dec_live_begin(psgpu_decode_s *d, int32_t max_frames, bool again, hipStream_t st);

int psgpu_decode_live_begin(psgpu_decode_t *d, int32_t max_frames, void *stream)
{
    return dec_live_begin(d, max_frames, false, (hipStream_t)stream);
}

// The utterance in progress begins AGAIN -- with room for max_frames frames; the caller feeds its frames from the first one -- from the
// session state it began with the first time: the seed lists and the multiplexed channels' ssids, both of which the steps so far have
// moved on (the seed slot follows the frames scored, the ssids are written back by every search).
int psgpu_decode_live_restart(psgpu_decode_t *d, int32_t max_frames, void *stream)
{
    PSGPU_REQUIRE(d && d->live && !d->streams, "psgpu_decode_live_restart: no live utterance (psgpu_decode_live_begin)");
    return dec_live_begin(d, max_frames, true, (hipStream_t)stream);
}

static int dec_live_begin(psgpu_decode_s *d, int32_t max_frames, bool again, hipStream_t st)
{
    PSGPU_REQUIRE(d && max_frames > 0, "psgpu_decode_live_begin: bad argument");
    PSGPU_REQUIRE(d->session, "psgpu_decode_live_begin: a live utterance is one decoder's (psgpu_decode_session first)");
    PSGPU_REQUIRE(!d->want_lists, "psgpu_decode_live_begin: a live utterance keeps its score rows (not with psgpu_decode_score_mode lists)");
    int rc;
    d->lists = false;                                    // (PSGPU_DECODE_LISTS: not for a live utterance; the next batch call picks its mode again)
    if ((rc = dec_settle_fe_ahead(d, st))) return rc;
    if ((rc = dec_session_buffers(d))) return rc;
    if ((rc = dec_grow(d, 1, (size_t)max_frames, (size_t)max_frames, st))) return rc;
    if (!d->d_pl_carry) {
        if ((rc = dec_alloc((void **)&d->d_seed0, (size_t)std::max(1, d->n_chain * d->topn)))
            || (rc = dec_alloc((void **)&d->d_lseed[0], (size_t)std::max(1, d->n_chain * d->topn)))
            || (rc = dec_alloc((void **)&d->d_lseed[1], (size_t)std::max(1, d->n_chain * d->topn)))
            || (rc = dec_alloc((void **)&d->d_off1, 16)) || (rc = dec_alloc((void **)&d->d_pl_carry, 4 * (size_t)psgpu_phone_loop_carry_words())))
            return rc;
    }
    d->streams = false;
    const size_t seed_bytes = (size_t)std::max(1, d->n_chain * d->topn);
    if (!again) {
        d->live_chained = d->sess_started; d->live_mpx_copied = false; d->live_searched = 0;
        d->seed0_valid = d->seed_valid;
        if (d->seed_valid) PSGPU_HIP(hipMemcpyAsync(d->d_seed0, d->d_seed, seed_bytes, hipMemcpyDeviceToDevice, st));
    }
    else {
        // (live_chained, and d_mpx_in once the first search has copied it, still hold what the utterance began from)
        d->seed_valid = d->seed0_valid;
        if (d->seed0_valid) PSGPU_HIP(hipMemcpyAsync(d->d_seed, d->d_seed0, seed_bytes, hipMemcpyDeviceToDevice, st));
    }
    d->live = true; d->live_ok = false; d->live_cap = max_frames; d->live_T = 0; d->live_S = 0; d->live_mode_next = 0;
    d->lseed_cur = 0;
    d->n_utt = 1; d->total = 0; d->max_frames = max_frames; d->searched = false; d->pass2 = false; d->first_called = true;
    d->bp_cap = (int32_t)d->cap_bp; d->bss_cap = (int32_t)d->cap_bss;
    d->frame_off.assign(2, 0);
    d->ev_valid = false;
    if (d->cfg.fe) psgpu_fe_offsets_dirty(d->cfg.fe);
    PSGPU_HIP(hipMemsetAsync(d->d_res, 0, 4 * 8, st));
    PSGPU_HIP(hipMemsetAsync(d->d_hn, 0, 4 * 4, st));
    return PSGPU_OK;
}

int psgpu_decode_live_step(psgpu_decode_t *d, const float *feat, int32_t n_new, int32_t lag, void *stream)
{
    PSGPU_REQUIRE(d && d->live && !d->streams, "psgpu_decode_live_step: no live utterance (psgpu_decode_live_begin)");
    PSGPU_REQUIRE(n_new >= 0 && lag >= 0 && (n_new == 0 || feat), "psgpu_decode_live_step: bad argument");
    // (a frame is searched with the look-ahead penalties of frame f + pl_window, phone_loop_search.c:302-340 / ngram_search_fwdtree.c:1453-1495:
    //  a search that stops fewer than pl_window frames short of the frames scored would take a clamped, wrong one for its last frames, and a
    //  resumed step never goes back to them)
    PSGPU_REQUIRE(lag == 0 || lag >= d->cfg.pl_window, "psgpu_decode_live_step: lag %d: 0 (the utterance's last step) or at least the look-ahead "
                  "window (%d frames)", lag, d->cfg.pl_window);
    PSGPU_REQUIRE(d->live_T + n_new <= d->live_cap, "psgpu_decode_live_step: %d frames exceed the live utterance's capacity of %d "
                  "(psgpu_decode_live_begin with a larger one, then the utterance's frames again)", d->live_T + n_new, d->live_cap);
    hipStream_t st = (hipStream_t)stream;
    int rc;
    const int t0 = d->live_T;
    if (n_new > 0) {
        const int32_t off1[3] = { 0, n_new, t0 }, off[2] = { 0, t0 + n_new };
        PSGPU_HIP(hipMemcpyAsync(d->d_off1, off1, 12, hipMemcpyHostToDevice, st));
        PSGPU_HIP(hipMemcpyAsync(d->d_off, off, 8, hipMemcpyHostToDevice, st));
        PSGPU_HIP(hipMemcpyAsync(d->d_feat + (size_t)t0 * d->veclen, feat, 4 * (size_t)n_new * d->veclen, hipMemcpyHostToDevice, st));
        PSGPU_HIP(hipStreamSynchronize(st));             // (feat and the offsets are the caller's / this frame's)
        const float *const f = d->d_feat + (size_t)t0 * d->veclen;
        int16_t *const rows = d->d_rows + (size_t)t0 * d->n_sen;
        if (d->kind == PSGPU_SCORER_PTM) {
            // this step's first frame starts from the lists of the frame before (ptm_mgau_frame_eval copies them, ptm_mgau.c:425-441):
            // the previous step's carry-out; the utterance's first frame from the session's seed, as in dec_from_feat
            const uint8_t *const seed_in = t0 > 0 ? d->d_lseed[d->lseed_cur] : ((d->live_chained && d->seed_valid) ? d->d_seed : nullptr);
            uint8_t *const seed_out = d->d_lseed[d->lseed_cur ^ 1];
            if ((rc = psgpu_ptm_score_batch_dev(d->cfg.model, f, d->d_off1, 1, n_new, seed_in, seed_out, d->d_tsc, d->d_tcw, rows, d->d_best + t0,
                                                d->compall ? 0u : PSGPU_PTM_RAW_SCORES, st)))
                return rc;
            d->lseed_cur ^= 1;
            // the next utterance's seed: the lists of the last frame ts of the utterance so far with ts % H == H - 1 (dec_from_feat)
            const int H = d->cfg.pl_window + 2;
            int ts = t0 + n_new - 1;
            while (ts >= t0 && ts % H != H - 1) --ts;
            if (ts >= t0) {
                PSGPU_HIP(hipMemcpy2DAsync(d->d_seed, (size_t)d->topn, d->d_tcw + (size_t)(ts - t0) * d->topn, (size_t)n_new * d->topn, (size_t)d->topn,
                                           (size_t)d->n_chain, hipMemcpyDeviceToDevice, st));
                d->seed_valid = true;
            }
        }
        else if (d->kind == PSGPU_SCORER_MS)
            rc = d->compall ? psgpu_ms_score_batch_dev((psgpu_ms_model_t *)d->cfg.scorer, f, n_new, d->d_ms_id, d->d_ms_dist, rows, st)
                            : psgpu_ms_score_batch_raw_dev((psgpu_ms_model_t *)d->cfg.scorer, f, n_new, d->d_ms_id, d->d_ms_dist, rows, st);
        else {
            // the semi-continuous scorer: as the PTM scorer above, the frames numbered from t0 (d_off1[2])
            const int H = d->cfg.pl_window + 2;
            const uint8_t *const seed_in = t0 > 0 ? d->d_lseed[d->lseed_cur] : ((d->live_chained && d->seed_valid) ? d->d_seed : nullptr);
            if ((rc = psgpu_semi_score_batch_carry_dev((psgpu_semi_model_t *)d->cfg.scorer, f, d->d_off1, 1, n_new, seed_in, d->d_lseed[d->lseed_cur ^ 1],
                                                       d->d_seed_tmp, H, d->d_off1 + 2, rows, st)))
                return rc;
            d->lseed_cur ^= 1;
            int ts = t0 + n_new - 1;
            while (ts >= t0 && ts % H != H - 1) --ts;
            if (ts >= t0) {
                PSGPU_HIP(hipMemcpyAsync(d->d_seed, d->d_seed_tmp, (size_t)d->n_chain * d->topn, hipMemcpyDeviceToDevice, st));
                d->seed_valid = true;
            }
        }
        if (rc) return rc;
        if ((rc = psgpu_phone_loop_run_carry_dev(d->cfg.ctx, &d->cfg.pl, d->d_ssid, d->d_tmatid, d->raw_flag == 3 ? nullptr : d->d_ci,
                                                 d->raw_flag == 3 ? 0 : d->cfg.n_ci_list, rows, d->n_sen, nullptr, d->d_off1, 1, n_new,
                                                 d->d_pen + (size_t)t0 * d->n_ci, d->d_pl_carry, t0 > 0, st)))
            return rc;
        d->live_T += n_new;
    }
    const int T = d->live_T;
    d->total = T; d->frame_off[1] = T; d->searched = false; d->pass2 = false;
    if (T == 0) return PSGPU_OK;                         // (nothing yet: the empty records of live_begin stand)
    if (d->live_chained && !d->live_mpx_copied) {        // (what this utterance's search starts from: a repeated search starts from it again)
        PSGPU_HIP(hipMemcpyAsync(d->d_mpx_in, d->d_mpx, 4 * (size_t)psgpu_fwdtree_n_mpx_channels(d->cfg.ft) * d->n_emit, hipMemcpyDeviceToDevice, st));
        d->live_mpx_copied = true;
    }
    // the search: up to `lag` frames short of the frames scored (the whole utterance when lag = 0, unless it is shorter than the
    // look-ahead window: see the kernel), from where the previous step's search stopped
    const int S = lag > 0 ? std::max(T - lag, 0) : (T < d->cfg.pl_window ? 0 : T);
    const int mode = dec_live_mode(d, d->live_ok && d->live_S <= S);
    d->live_mode_next = mode;
    d->live_searched += S - ((mode & PSGPU_SEARCH_RESUME) ? d->live_S : 0);
    d->last_chained = d->live_chained; d->last_sess = true; d->last_lag = lag; d->lag_next = 0;
    if ((rc = dec_search(d, 1, (size_t)T, (size_t)d->live_cap, st))) { d->live_ok = false; return rc; }
    d->live_S = S; d->live_ok = (mode & PSGPU_SEARCH_KEEP) != 0;
    d->sess_started = true;
    return PSGPU_OK;
}

int64_t psgpu_decode_live_frames_searched(const psgpu_decode_t *d) { return d ? (d->streams ? d->ls_searched : d->live_searched) : 0; }

// ---- many utterances in progress ---------------------------------------------------------------------------------------------------
int psgpu_decode_streams_begin(psgpu_decode_t *d, int32_t n_streams, int32_t max_frames, int32_t max_step_frames, void *stream)
{
    PSGPU_REQUIRE(d && n_streams > 0 && max_frames > 0 && max_step_frames > 0, "psgpu_decode_streams_begin: bad argument");
    PSGPU_REQUIRE(!d->want_lists && (d->n_sen & 1) == 0, "psgpu_decode_streams_begin: streams keep score rows (an even number of senones; not with "
                  "psgpu_decode_score_mode lists)");
    hipStream_t st = (hipStream_t)stream;
    int rc;
    d->lists = false; d->live = true;                    // (score rows; dec_pick_mode keeps them while the streams are in progress)
    if ((rc = dec_settle_fe_ahead(d, st))) return rc;
    // a stream's search cannot be repeated with larger arrays (its rows are gone once searched): the slab layouts' capacities at their
    // ends from the start, so that no stream ends with status 4 / 5 / 6
    if ((rc = psgpu_fwdtree_full_capacity(d->cfg.ft))) return rc;
    const size_t step_total = (size_t)n_streams * max_step_frames;
    if ((rc = dec_grow(d, (size_t)n_streams, step_total, (size_t)max_frames, st))) return rc;
    PSGPU_HIP(hipStreamSynchronize(st));
    const int lag = d->cfg.pl_window;
    const size_t wrows = (size_t)n_streams * ((size_t)max_step_frames + lag + 1), per = (size_t)std::max(1, d->n_chain * d->topn);
    if (wrows > d->win_rows || n_streams > d->ls_n) {
        DFREE(d->d_win[0]); DFREE(d->d_win[1]); DFREE(d->d_wpen[0]); DFREE(d->d_wpen[1]); DFREE(d->d_ls); DFREE(d->d_sseed[0]); DFREE(d->d_sseed[1]);
        DFREE(d->d_splc); DFREE(d->d_sslot); DFREE(d->d_smpx_in); DFREE(d->d_smpx_out);
        d->win_rows = 0;
        for (int k = 0; k < 2; ++k)
            if ((rc = dec_alloc((void **)&d->d_win[k], 2 * wrows * d->n_sen + 64)) || (rc = dec_alloc((void **)&d->d_wpen[k], 4 * wrows * d->n_ci))
                || (rc = dec_alloc((void **)&d->d_sseed[k], (size_t)n_streams * per)))
                return rc;
        DFREE(d->d_sslot); DFREE(d->d_smpx_in); DFREE(d->d_smpx_out);
        const size_t mpxw = (size_t)std::max(1, psgpu_fwdtree_n_mpx_channels(d->cfg.ft)) * d->n_emit;
        if ((rc = dec_alloc((void **)&d->d_ls, 4 * (size_t)(12 * n_streams + 8)))
            || (rc = dec_alloc((void **)&d->d_splc, 4 * (size_t)n_streams * psgpu_phone_loop_carry_words()))
            || (rc = dec_alloc((void **)&d->d_sslot, (size_t)n_streams * per))
            || (rc = dec_alloc((void **)&d->d_smpx_in, 4 * (size_t)n_streams * mpxw)) || (rc = dec_alloc((void **)&d->d_smpx_out, 4 * (size_t)n_streams * mpxw)))
            return rc;
        d->win_rows = wrows;
    }
    PSGPU_HIP(hipMemsetAsync(d->d_splc, 0, 4 * (size_t)n_streams * psgpu_phone_loop_carry_words(), st));
    {   // a new scorer's lists: codeword = rank (ptm_mgau.c:790-793)
        std::vector<uint8_t> seed((size_t)n_streams * per);
        for (size_t i = 0; i < seed.size(); ++i) seed[i] = (uint8_t)(d->topn > 0 ? (i % per) % d->topn : 0);
        PSGPU_HIP(hipMemcpyAsync(d->d_sseed[0], seed.data(), seed.size(), hipMemcpyHostToDevice, st));
        PSGPU_HIP(hipStreamSynchronize(st));
    }
    d->streams = true; d->ls_first = true; d->ls_n = n_streams; d->ls_cap = max_frames; d->ls_step = max_step_frames; d->ls_lag = lag; d->ls_cur = 0; d->ls_wcur = 0;
    d->ls_searched = 0;
    d->ls_T.assign(n_streams, 0); d->ls_S.assign(n_streams, 0); d->ls_wbase.assign(n_streams, 0); d->ls_woff.assign(n_streams, 0);
    d->ls_fresh.assign(n_streams, 0); d->ls_slot.assign(n_streams, 0); d->ls_mpx.assign(n_streams, 0);
    d->n_utt = n_streams; d->total = 0; d->max_frames = max_frames; d->searched = false; d->pass2 = false; d->first_called = true;
    d->bp_cap = (int32_t)d->cap_bp; d->bss_cap = (int32_t)d->cap_bss;
    d->frame_off.assign((size_t)n_streams + 1, 0);
    d->ev_valid = false;
    if (d->cfg.fe) psgpu_fe_offsets_dirty(d->cfg.fe);
    PSGPU_HIP(hipMemsetAsync(d->d_res, 0, 4 * (size_t)n_streams * 8, st));
    PSGPU_HIP(hipMemsetAsync(d->d_hn, 0, 4 * (size_t)n_streams * 4, st));
    return PSGPU_OK;
}

int psgpu_decode_streams_restart(psgpu_decode_t *d, int32_t u, void *stream)
{
    PSGPU_REQUIRE(d && d->streams && u >= 0 && u < d->ls_n, "psgpu_decode_streams_restart: bad argument");
    hipStream_t st = (hipStream_t)stream;
    int rc;
    d->ls_T[u] = 0; d->ls_S[u] = 0; d->ls_wbase[u] = 0; d->ls_woff[u] = 0; d->ls_fresh[u] = 1;
    d->ls_slot[u] = 0; d->ls_mpx[u] = 0;                  // (a new decoder: nothing inherited)
    if (d->pcm_streams && (rc = dec_pcm_stream_reset(d, u, true, st))) return rc;
    if (!d->ls_first && (rc = psgpu_fwdtree_search_restart(d->cfg.ft, u, st))) return rc;
    if ((rc = psgpu_phone_loop_carry_restart(d->d_splc, u, st))) return rc;
    const size_t per = (size_t)std::max(1, d->n_chain * d->topn);
    std::vector<uint8_t> seed(per);
    for (size_t i = 0; i < per; ++i) seed[i] = (uint8_t)(d->topn > 0 ? i % d->topn : 0);
    PSGPU_HIP(hipMemcpyAsync(d->d_sseed[d->ls_cur] + (size_t)u * per, seed.data(), per, hipMemcpyHostToDevice, st));
    PSGPU_HIP(hipStreamSynchronize(st));
    return PSGPU_OK;
}

// Stream u's decoder goes on to its NEXT utterance: what ps_start_utt leaves in place (the scorer's ring slot that seeds the first
// frame, the multiplexed channels' ssids -- psgpu_decode_session's carry-over, per stream), everything else afresh.
int psgpu_decode_streams_next_utt(psgpu_decode_t *d, int32_t u, void *stream)
{
    PSGPU_REQUIRE(d && d->streams && u >= 0 && u < d->ls_n, "psgpu_decode_streams_next_utt: bad argument");
    PSGPU_REQUIRE(!d->ls_first, "psgpu_decode_streams_next_utt: the stream has had no utterance yet");
    hipStream_t st = (hipStream_t)stream;
    int rc;
    const size_t per = (size_t)std::max(1, d->n_chain * d->topn), mpxw = (size_t)std::max(1, psgpu_fwdtree_n_mpx_channels(d->cfg.ft)) * d->n_emit;
    const uint8_t slot = d->ls_slot[u];
    std::vector<float> fstate;
    LiveSim keep;
    int32_t undef = 0;
    if (d->pcm_streams) {
        // (the decoder's front half goes on: noise tracker, running cepstral mean and the feature ring stay -- fe_start_utt /
        //  acmod_start_utt reset the overflow buffer, the prior and the buffers' counters only)
        fstate.resize(d->pc_init.size());
        PSGPU_HIP(hipMemcpyAsync(fstate.data(), d->d_pfeat + (size_t)u * fstate.size(), 4 * fstate.size(), hipMemcpyDeviceToHost, st));
        PSGPU_HIP(hipMemcpyAsync(&undef, d->d_pundef + u, 4, hipMemcpyDeviceToHost, st));
        PSGPU_HIP(hipStreamSynchronize(st));
        keep = d->pc_sim[u];
    }
    if ((rc = psgpu_decode_streams_restart(d, u, stream))) return rc;       // (search, phone loop, window; the lists: a new scorer's)
    if (d->pcm_streams) {
        d->pc_sim[u].nbuf = keep.nbuf; d->pc_sim[u].FA = keep.FA; d->pc_sim[u].live_seen = keep.live_seen;
        PSGPU_HIP(hipMemcpyAsync(d->d_pfeat + (size_t)u * fstate.size(), fstate.data(), 4 * fstate.size(), hipMemcpyHostToDevice, st));
        PSGPU_HIP(hipMemcpyAsync(d->d_pundef + u, &undef, 4, hipMemcpyHostToDevice, st));
        PSGPU_HIP(hipStreamSynchronize(st));
    }
    d->ls_slot[u] = slot; d->ls_mpx[u] = 1;
    PSGPU_HIP(hipMemcpyAsync(d->d_smpx_in + (size_t)u * mpxw, d->d_smpx_out + (size_t)u * mpxw, 4 * mpxw, hipMemcpyDeviceToDevice, st));
    if (slot) PSGPU_HIP(hipMemcpyAsync(d->d_sseed[d->ls_cur] + (size_t)u * per, d->d_sslot + (size_t)u * per, per, hipMemcpyDeviceToDevice, st));
    return PSGPU_OK;
}

static int dec_streams_step_core(psgpu_decode_t *d, const float *feat, bool feat_on_device, const int32_t *n_new, const uint8_t *final_flags, void *stream);

int psgpu_decode_streams_step(psgpu_decode_t *d, const float *feat, const int32_t *n_new, const uint8_t *final_flags, void *stream)
{
    PSGPU_REQUIRE(d && d->streams && n_new, "psgpu_decode_streams_step: no streams (psgpu_decode_streams_begin) / NULL argument");
    PSGPU_REQUIRE(!d->pcm_streams, "psgpu_decode_streams_step: these streams are fed with audio (psgpu_decode_streams_step_pcm)");
    return dec_streams_step_core(d, feat, false, n_new, final_flags, stream);
}

static int dec_streams_step_core(psgpu_decode_t *d, const float *feat, bool feat_on_device, const int32_t *n_new, const uint8_t *final_flags, void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    const int n = d->ls_n, lag = d->ls_lag;
    int rc;
    // the step's tables
    std::vector<int32_t> &h = d->ls_h;
    h.assign((size_t)12 * n + 8, 0);
    int32_t *const off1 = h.data(), *const uoff = off1 + n + 1, *const ext = uoff + n + 1, *const map = ext + 3 * n, *const fbase = map + 5 * n;
    size_t total = 0, wtot = 0;
    int64_t searched = 0;
    for (int u = 0; u < n; ++u) {
        PSGPU_REQUIRE(n_new[u] >= 0 && n_new[u] <= d->ls_step && d->ls_T[u] + n_new[u] <= d->ls_cap,
                      "psgpu_decode_streams_step: stream %d: %d new frames (at most %d a step, %d an utterance)", u, n_new[u], d->ls_step, d->ls_cap);
        const int T0 = d->ls_T[u], S0 = d->ls_S[u], T = T0 + n_new[u], keep = T0 - S0;
        const bool fin = final_flags && final_flags[u];
        const int S = fin ? (T < d->cfg.pl_window ? S0 : T) : std::max(T - lag, S0);
        off1[u] = (int32_t)total; total += (size_t)n_new[u];
        fbase[u] = T0;
        map[5 * u] = d->ls_woff[u] + (S0 - d->ls_wbase[u]); map[5 * u + 1] = keep; map[5 * u + 2] = off1[u]; map[5 * u + 3] = n_new[u];
        map[5 * u + 4] = (int32_t)wtot;
        uoff[u] = (int32_t)wtot - S0;                    // frame f's row: (uoff + f) -- the window starts at frame S0
        ext[3 * u] = T; ext[3 * u + 1] = S; ext[3 * u + 2] = d->ls_mpx[u];
        searched += S - S0;
        wtot += (size_t)keep + n_new[u];
    }
    off1[n] = (int32_t)total; uoff[n] = 0;
    PSGPU_REQUIRE(wtot <= d->win_rows, "psgpu_decode_streams_step: %zu rows exceed the window buffer (%zu)", wtot, d->win_rows);
    PSGPU_REQUIRE(total == 0 || feat || feat_on_device, "psgpu_decode_streams_step: NULL features");
    PSGPU_HIP(hipMemcpyAsync(d->d_ls, h.data(), 4 * h.size(), hipMemcpyHostToDevice, st));
    if (total && !feat_on_device) PSGPU_HIP(hipMemcpyAsync(d->d_feat, feat, 4 * total * d->veclen, hipMemcpyHostToDevice, st));
    PSGPU_HIP(hipStreamSynchronize(st));                 // (feat is the caller's)
    const int32_t *const d_off1 = d->d_ls, *const d_uoff = d->d_ls + n + 1, *const d_ext = d_uoff + n + 1, *const d_map = d_ext + 3 * n,
                  *const d_fbase = d_map + 5 * n;
    const size_t per = (size_t)std::max(1, d->n_chain * d->topn);
    if (total) {
        if (d->kind == PSGPU_SCORER_PTM || d->kind == PSGPU_SCORER_SEMI) {
            const uint8_t *const seed_in = d->d_sseed[d->ls_cur];
            uint8_t *const seed_out = d->d_sseed[d->ls_cur ^ 1];
            if (d->kind == PSGPU_SCORER_PTM)
                rc = psgpu_ptm_score_batch_dev(d->cfg.model, d->d_feat, d_off1, n, (int32_t)total, seed_in, seed_out, d->d_tsc, d->d_tcw, d->d_rows,
                                               d->d_best, d->compall ? 0u : PSGPU_PTM_RAW_SCORES, st);
            else
                rc = psgpu_semi_score_batch_carry_dev((psgpu_semi_model_t *)d->cfg.scorer, d->d_feat, d_off1, n, (int32_t)total, seed_in, seed_out,
                                                      d->d_sslot, d->cfg.pl_window + 2, d_fbase, d->d_rows, st);
            if (rc) return rc;
            if (d->kind == PSGPU_SCORER_PTM) {
                hipLaunchKernelGGL(dec_slot_kernel, dim3((unsigned)n), dim3(64), 0, st, d->d_tcw, d_off1, d_fbase, (int32_t)total, d->n_chain, d->topn,
                                   d->cfg.pl_window + 2, d->d_sslot);
                PSGPU_HIP(hipGetLastError());
            }
            for (int u = 0; u < n; ++u) {                // (the ring slot was written if the step held a frame ts with ts % H == H - 1)
                const int H = d->cfg.pl_window + 2, a = d->ls_T[u];
                int ts = a + n_new[u] - 1;
                while (ts >= a && ts % H != H - 1) --ts;
                if (ts >= a) d->ls_slot[u] = 1;
            }
            hipLaunchKernelGGL(dec_seed_keep_kernel, dim3((unsigned)n), dim3(64), 0, st, d_off1, n, (int32_t)per, seed_in, seed_out);
            PSGPU_HIP(hipGetLastError());
            d->ls_cur ^= 1;
        }
        else
            rc = d->compall ? psgpu_ms_score_batch_dev((psgpu_ms_model_t *)d->cfg.scorer, d->d_feat, (int32_t)total, d->d_ms_id, d->d_ms_dist, d->d_rows, st)
                            : psgpu_ms_score_batch_raw_dev((psgpu_ms_model_t *)d->cfg.scorer, d->d_feat, (int32_t)total, d->d_ms_id, d->d_ms_dist, d->d_rows, st);
        if (rc) return rc;
        if ((rc = psgpu_phone_loop_run_carry_dev(d->cfg.ctx, &d->cfg.pl, d->d_ssid, d->d_tmatid, d->raw_flag == 3 ? nullptr : d->d_ci,
                                                 d->raw_flag == 3 ? 0 : d->cfg.n_ci_list, d->d_rows, d->n_sen, nullptr, d_off1, n, (int32_t)total,
                                                 d->d_pen, d->d_splc, 1, st)))
            return rc;
    }
    // the windows: what the search has not reached yet + the step's rows, into the other window buffer
    const int wo = d->ls_wcur, wn = wo ^ 1;
    const int32_t max_rows = d->ls_step + lag + 1;
    hipLaunchKernelGGL(dec_window_kernel, dim3((unsigned)((size_t)n * max_rows)), dim3(256), 0, st, d_map, n, max_rows, d->n_sen / 2, d->n_ci,
                       reinterpret_cast<const uint32_t *>(d->d_win[wo]), reinterpret_cast<const uint32_t *>(d->d_rows),
                       reinterpret_cast<uint32_t *>(d->d_win[wn]), d->d_wpen[wo], d->d_pen, d->d_wpen[wn]);
    PSGPU_HIP(hipGetLastError());
    d->ls_wcur = wn;
    // the searches go on (a stream that starts afresh: its saved state was invalidated by psgpu_decode_streams_restart)
    if ((rc = psgpu_fwdtree_hyp_out(d->cfg.ft, d->d_hyp, d->d_hn, d->max_words))) return rc;
    if ((rc = psgpu_fwdtree_search_lag(d->cfg.ft, 0))) return rc;
    if ((rc = psgpu_fwdtree_search_streams(d->cfg.ft, d_ext))) return rc;
    if ((rc = psgpu_fwdtree_search_resume(d->cfg.ft, PSGPU_SEARCH_KEEP | (d->ls_first ? 0 : PSGPU_SEARCH_RESUME)))) return rc;
    if ((rc = psgpu_fwdtree_search_session_dev(d->cfg.ft, d->d_win[wn], d->n_sen, d->d_wpen[wn], d_uoff, n, d->ls_cap, d->bp_cap, d->bss_cap,
                                               d->d_bp, d->d_bss, d->d_idx, d->d_step, d->d_res, d->raw_flag, d->cfg.pl_window, d->d_w1,
                                               d->d_smpx_in, d->d_smpx_out, st)))
        return rc;
    d->ls_first = false; d->searched = true; d->pass2 = false; d->last_lag = 0; d->last_chained = false; d->last_sess = false;
    d->ls_searched += searched;
    size_t acc = 0;
    for (int u = 0; u < n; ++u) {
        d->ls_wbase[u] = d->ls_S[u]; d->ls_woff[u] = map[5 * u + 4];
        d->ls_T[u] = ext[3 * u]; d->ls_S[u] = ext[3 * u + 1]; d->ls_fresh[u] = 0;
        d->frame_off[u] = (int32_t)acc; acc += (size_t)d->ls_T[u];
    }
    d->frame_off[n] = (int32_t)acc; d->total = (int32_t)std::min<size_t>(acc, 0x7fffffff);
    return PSGPU_OK;
}

// ---- the streams fed with audio ----------------------------------------------------------------------------------------------------
// assemble: work = per stream [prior | carried samples | the step's new samples]; desc [n][4] = {prior slot's index in work, carried, the new
// samples' offset in the step's buffer, new}
__global__ __launch_bounds__(256)
void dec_pcm_assemble_kernel(const int32_t *__restrict__ desc, const int16_t *__restrict__ carry, int32_t slots, const int16_t *__restrict__ pcm,
                             int16_t *__restrict__ work)
{
    const int u = blockIdx.y;
    const int32_t w0 = desc[4 * u], nc = desc[4 * u + 1], p0 = desc[4 * u + 2], nn = desc[4 * u + 3];
    for (int i = blockIdx.x * 256 + threadIdx.x; i < 1 + nc + nn; i += gridDim.x * 256)
        work[w0 + i] = i <= nc ? carry[(size_t)u * slots + i] : pcm[p0 + (i - 1 - nc)];
}
// ... and what the step leaves unframed goes back: desc2 [n][3] = {prior slot's index in work, samples framed away, samples left}; an
// utterance that ended leaves nothing and a zero prior (fe_start_utt, fe_interface.c:321-331)
__global__ __launch_bounds__(256)
void dec_pcm_carry_kernel(const int32_t *__restrict__ desc2, const int16_t *__restrict__ work, int32_t slots, int16_t *__restrict__ carry)
{
    const int u = blockIdx.x;
    const int32_t w0 = desc2[3 * u], q = desc2[3 * u + 1], keep = desc2[3 * u + 2];
    for (int i = threadIdx.x; i < 1 + (keep > 0 ? keep : 0); i += 256)
        carry[(size_t)u * slots + i] = keep < 0 ? (int16_t)0 : work[w0 + q + i];
}

static int dec_pcm_stream_reset(psgpu_decode_s *d, int u, bool new_decoder, hipStream_t st)
{
    d->pc_sim[u].start();
    d->pc_ncarry[u] = 0;
    PSGPU_HIP(hipMemsetAsync(d->d_pcarry + (size_t)u * d->pc_slots, 0, 2, st));             // (the prior)
    if (new_decoder) {                                   // ps_start_stream + a new decoder's feat_t / cmn_t
        const int one = 1;
        d->pc_sim[u].nbuf = 0; d->pc_sim[u].FA = d->pc_sim[u].FA0; d->pc_sim[u].live_seen = false;
        PSGPU_HIP(hipMemcpyAsync(d->d_pundef + u, &one, 4, hipMemcpyHostToDevice, st));
        PSGPU_HIP(hipMemcpyAsync(d->d_pfeat + (size_t)u * d->pc_init.size(), d->pc_init.data(), 4 * d->pc_init.size(), hipMemcpyHostToDevice, st));
        PSGPU_HIP(hipStreamSynchronize(st));
    }
    return PSGPU_OK;
}

int psgpu_decode_streams_pcm_begin(psgpu_decode_t *d, int32_t n_streams, int32_t max_frames, int32_t max_step_frames, const float *cmninit,
                                   int32_t n_cmninit, int32_t grow_feat, void *stream)
{
    PSGPU_REQUIRE(d && d->cfg.fe, "psgpu_decode_streams_pcm_begin: the pipeline has no front end");
    PSGPU_REQUIRE(d->veclen == 3 * d->cepsize, "psgpu_decode_streams_pcm_begin: the live feature computation is 1s_c_d_dd's (feat.c:579-622)");
    int rc;
    if ((rc = psgpu_decode_streams_begin(d, n_streams, max_frames, max_step_frames, stream))) return rc;
    hipStream_t st = (hipStream_t)stream;
    d->pc_fs = psgpu_fe_frame_size(d->cfg.fe); d->pc_sh = psgpu_fe_frame_shift(d->cfg.fe);
    d->pc_slots = d->pc_fs + 8;
    const int words = psgpu_feat_live_state_words(d->cepsize);
    d->pc_init.assign((size_t)words, 0.0f);
    if ((rc = psgpu_feat_live_state_init(d->pc_init.data(), d->cepsize, cmninit, n_cmninit))) return rc;
    DFREE(d->d_pcarry); DFREE(d->d_pnoise); DFREE(d->d_pundef); DFREE(d->d_pdesc); DFREE(d->d_pfoff); DFREE(d->d_pfeat);
    const int n_filt = 64;                               // (the noise tracker's state: at most 64 mel channels, psgpu_fe_create)
    if ((rc = dec_alloc((void **)&d->d_pcarry, 2 * (size_t)n_streams * d->pc_slots)) || (rc = dec_alloc((void **)&d->d_pnoise, 8 * (size_t)n_streams * 4 * n_filt))
        || (rc = dec_alloc((void **)&d->d_pundef, 4 * (size_t)n_streams)) || (rc = dec_alloc((void **)&d->d_pdesc, 4 * (size_t)n_streams * 16))
        || (rc = dec_alloc((void **)&d->d_pfoff, 4 * ((size_t)n_streams + 1) * 4)) || (rc = dec_alloc((void **)&d->d_pfeat, 4 * (size_t)n_streams * words)))
        return rc;
    d->pc_sim.assign((size_t)n_streams, LiveSim());
    d->pc_ncarry.assign((size_t)n_streams, 0);
    d->pcm_streams = true;
    for (int u = 0; u < n_streams; ++u) {
        d->pc_sim[u].setup(d->pc_fs, d->pc_sh, 3, d->cfg.pl_window, grow_feat != 0);
        d->pc_sim[u].ops = &d->pc_ops;
        if ((rc = dec_pcm_stream_reset(d, u, true, st))) return rc;
    }
    return PSGPU_OK;
}

int psgpu_decode_streams_step_pcm(psgpu_decode_t *d, const int16_t *pcm, const int64_t *n_samples, const uint8_t *final_flags, int32_t *n_new_out,
                                  void *stream)
{
    PSGPU_REQUIRE(d && d->streams && d->pcm_streams && n_samples, "psgpu_decode_streams_step_pcm: no audio streams (psgpu_decode_streams_pcm_begin) / NULL argument");
    hipStream_t st = (hipStream_t)stream;
    const int n = d->ls_n;
    int rc;
    // the reference's counters for the step: frames the front end makes, the pieces, the feature frames they release
    std::vector<int32_t> &h = d->pc_h;
    h.assign((size_t)n * 12 + 8, 0);
    int32_t *const desc = h.data(), *const desc2 = desc + 4 * n, *const nfr = desc2 + 3 * n, *const op_off = nfr + n, *const n_new = op_off + n + 1;
    std::vector<int64_t> samp(2 * (size_t)n);
    d->pc_ops.clear();
    int64_t work = 0, pcm_total = 0, cep_total = 0, feat_total = 0;
    std::vector<int32_t> foff((size_t)n + 1, 0);
    for (int u = 0; u < n; ++u) {
        PSGPU_REQUIRE(n_samples[u] >= 0 && n_samples[u] < ((int64_t)1 << 30), "psgpu_decode_streams_step_pcm: stream %d: %lld samples", u, (long long)n_samples[u]);
        LiveSim &sim = d->pc_sim[u];
        const bool fin = final_flags && final_flags[u];
        PSGPU_REQUIRE(sim.state != 3 || (n_samples[u] == 0 && !fin), "psgpu_decode_streams_step_pcm: stream %d's utterance has ended (psgpu_decode_streams_next_utt / "
                      "_restart begins the next)", u);
        sim.made = 0; sim.feats = 0;
        op_off[u] = (int32_t)(d->pc_ops.size() / 2);
        const int nc = d->pc_ncarry[u];
        if (sim.state != 3) {
            if (n_samples[u] > 0) sim.process_raw(n_samples[u]);
            int tail = 0;
            if (fin) tail = sim.end();
            const int64_t avail = nc + n_samples[u];
            const int full = sim.made - (tail ? 1 : 0);
            const int64_t left = avail - (int64_t)full * d->pc_sh;
            // (an utterance's end without fe_end_utt -- acmod_end_utt with a full cepstrum buffer, acmod.c:428 -- leaves the samples to fe_start_utt)
            const int64_t expect = fin ? (tail ? tail : left) : sim.ov;
            PSGPU_REQUIRE(left >= 0 && left == expect && left < d->pc_slots - 1, "psgpu_decode_streams_step_pcm: stream %d: the sample counters "
                          "disagree (%lld left, %lld in the overflow buffer)", u, (long long)left, (long long)expect);
            desc2[3 * u + 1] = full * d->pc_sh; desc2[3 * u + 2] = fin ? -1 : (int32_t)left;
            d->pc_ncarry[u] = fin ? 0 : (int32_t)left;
        }
        else { desc2[3 * u + 1] = 0; desc2[3 * u + 2] = nc; }
        desc[4 * u] = (int32_t)work; desc[4 * u + 1] = nc; desc[4 * u + 2] = (int32_t)pcm_total; desc[4 * u + 3] = (int32_t)n_samples[u];
        desc2[3 * u] = (int32_t)work;
        samp[2 * u] = work + 1; samp[2 * u + 1] = nc + n_samples[u];
        work += 1 + nc + n_samples[u]; pcm_total += n_samples[u];
        PSGPU_REQUIRE(work < ((int64_t)1 << 31), "psgpu_decode_streams_step_pcm: more than 2^31 samples in a step");
        nfr[u] = sim.made; cep_total += sim.made;
        n_new[u] = sim.feats; foff[u] = (int32_t)feat_total; feat_total += sim.feats;
        PSGPU_REQUIRE(sim.feats <= d->ls_step, "psgpu_decode_streams_step_pcm: stream %d: %d feature frames in one step (at most %d: psgpu_decode_streams_pcm_begin)",
                      u, sim.feats, d->ls_step);
    }
    op_off[n] = (int32_t)(d->pc_ops.size() / 2); foff[n] = (int32_t)feat_total;
    if (n_new_out) memcpy(n_new_out, n_new, 4 * (size_t)n);
    PSGPU_REQUIRE(pcm_total == 0 || pcm, "psgpu_decode_streams_step_pcm: NULL audio");
    // buffers
    if ((size_t)work > d->pc_work_cap || !d->d_pwork) { DFREE(d->d_pwork); d->pc_work_cap = 0; if ((rc = dec_alloc((void **)&d->d_pwork, 2 * ((size_t)work + work / 2 + 64)))) return rc; d->pc_work_cap = (size_t)work + work / 2 + 64; }
    if ((size_t)pcm_total > d->pc_pcm_cap || !d->d_ppcm) { DFREE(d->d_ppcm); d->pc_pcm_cap = 0; if ((rc = dec_alloc((void **)&d->d_ppcm, 2 * ((size_t)pcm_total + pcm_total / 2 + 64)))) return rc; d->pc_pcm_cap = (size_t)pcm_total + pcm_total / 2 + 64; }
    if (d->pc_ops.size() + 2 > d->pc_ops_cap || !d->d_pops) { DFREE(d->d_pops); d->pc_ops_cap = 0; if ((rc = dec_alloc((void **)&d->d_pops, 4 * (2 * d->pc_ops.size() + 64)))) return rc; d->pc_ops_cap = 2 * d->pc_ops.size() + 64; }
    if ((size_t)cep_total > d->pc_cep_cap || !d->d_pcep) { DFREE(d->d_pcep); d->pc_cep_cap = 0; if ((rc = dec_alloc((void **)&d->d_pcep, 4 * ((size_t)cep_total * 2 + 64) * d->cepsize))) return rc; d->pc_cep_cap = (size_t)cep_total * 2 + 64; }
    PSGPU_HIP(hipMemcpyAsync(d->d_pdesc, h.data(), 4 * (size_t)(7 * n), hipMemcpyHostToDevice, st));
    if (pcm_total) PSGPU_HIP(hipMemcpyAsync(d->d_ppcm, pcm, 2 * (size_t)pcm_total, hipMemcpyHostToDevice, st));
    if (!d->pc_ops.empty()) PSGPU_HIP(hipMemcpyAsync(d->d_pops, d->pc_ops.data(), 4 * d->pc_ops.size(), hipMemcpyHostToDevice, st));
    // op offsets [n + 1], feature offsets [n + 1] (the cepstra's offsets come from the front end: d_pfoff[0 .. n])
    int32_t *const d_opoff = d->d_pfoff + (n + 1), *const d_featoff = d_opoff + (n + 1);
    PSGPU_HIP(hipMemcpyAsync(d_opoff, op_off, 4 * ((size_t)n + 1), hipMemcpyHostToDevice, st));
    PSGPU_HIP(hipMemcpyAsync(d_featoff, foff.data(), 4 * ((size_t)n + 1), hipMemcpyHostToDevice, st));
    PSGPU_HIP(hipStreamSynchronize(st));                 // (pcm, the vectors: the caller's / this call's)
    {
        const int64_t most = d->pc_slots + (pcm_total > 0 ? *std::max_element(n_samples, n_samples + n) : 0);
        hipLaunchKernelGGL(dec_pcm_assemble_kernel, dim3((unsigned)std::min<int64_t>((most + 255) / 256, 64), (unsigned)n), dim3(256), 0, st, d->d_pdesc, d->d_pcarry,
                           d->pc_slots, d->d_ppcm, d->d_pwork);
        PSGPU_HIP(hipGetLastError());
    }
    if ((rc = psgpu_fe_stream_step_dev(d->cfg.fe, d->d_pwork, samp.data(), nfr, n, d->d_pnoise, d->d_pundef, d->d_pcep, d->d_pfoff, st))) return rc;
    hipLaunchKernelGGL(dec_pcm_carry_kernel, dim3((unsigned)n), dim3(256), 0, st, d->d_pdesc + 4 * n, d->d_pwork, d->pc_slots, d->d_pcarry);
    PSGPU_HIP(hipGetLastError());
    // (the feature rows of the step go where psgpu_decode_streams_step puts the caller's: room for ls_step rows a stream, psgpu_decode_streams_begin)
    if ((rc = psgpu_feat_live_step_dev(d->d_pcep, d->d_pfoff, d->d_pops, d_opoff, d_featoff, n, d->cepsize, d->d_pfeat, d->d_feat, st))) return rc;
    return dec_streams_step_core(d, nullptr, true, n_new, final_flags, stream);
}

// the counters alone, for one utterance from its start: a host-only entry (no device is touched) -- a binding that wants to know what a
// live reference decoder does with a sequence of ps_process_raw calls (and the tests of LiveSim against the reference's dumps)
int psgpu_live_pieces(int32_t frame_size, int32_t frame_shift, int32_t window, int32_t pl_window, int32_t grow_feat, const int64_t *chunks,
                      int32_t n_chunks, int32_t final_, int32_t *ops_out, int32_t ops_cap, int32_t *n_ops, int32_t *n_cepstra, int32_t *n_feat_frames)
{
    PSGPU_REQUIRE(frame_size > 0 && frame_shift > 0 && frame_shift <= frame_size && window >= 0 && pl_window >= 0 && n_chunks >= 0 && (n_chunks == 0 || chunks),
                  "psgpu_live_pieces: bad argument");
    LiveSim sim;
    std::vector<int32_t> ops;
    sim.setup(frame_size, frame_shift, window, pl_window, grow_feat != 0);
    sim.ops = &ops;
    sim.start();
    for (int k = 0; k < n_chunks; ++k) { PSGPU_REQUIRE(chunks[k] >= 0, "psgpu_live_pieces: negative chunk"); if (chunks[k]) sim.process_raw(chunks[k]); }
    if (final_) sim.end();
    if (n_ops) *n_ops = (int32_t)(ops.size() / 2);
    if (n_cepstra) *n_cepstra = sim.made;
    if (n_feat_frames) *n_feat_frames = sim.feats;
    if (ops_out) {
        PSGPU_REQUIRE((size_t)ops_cap * 2 >= ops.size(), "psgpu_live_pieces: %zu ops, room for %d", ops.size() / 2, ops_cap);
        memcpy(ops_out, ops.data(), 4 * ops.size());
    }
    return PSGPU_OK;
}

int psgpu_decode_stage_timing(psgpu_decode_t *d, int32_t enable)
{
    PSGPU_REQUIRE(d, "psgpu_decode_stage_timing: NULL argument");
    if (enable && !d->ev[0])
        for (int i = 0; i < 7; ++i) PSGPU_HIP(hipEventCreate(&d->ev[i]));
    d->timing = enable != 0;
    d->ev_valid = false;
    return PSGPU_OK;
}

int psgpu_decode_last_stage_ms(psgpu_decode_t *d, float ms[6])
{
    PSGPU_REQUIRE(d && ms, "psgpu_decode_last_stage_ms: NULL argument");
    PSGPU_REQUIRE(d->ev_valid, "psgpu_decode_last_stage_ms: no timed psgpu_decode_first_pass_dev call yet");
    PSGPU_HIP(hipEventSynchronize(d->ev[6]));
    for (int i = 0; i < 4; ++i) PSGPU_HIP(hipEventElapsedTime(&ms[i], d->ev[i], d->ev[i + 1]));
    PSGPU_HIP(hipEventElapsedTime(&ms[4], d->ev[5], d->ev[6]));          // the search kernel (its last step is the backtrace)
    PSGPU_HIP(hipEventElapsedTime(&ms[5], d->ev[4], d->ev[5]));          // waiting for another object's search (psgpu_decode_search_after)
    return PSGPU_OK;
}

int psgpu_decode_view(const psgpu_decode_t *d, psgpu_decode_view_t *v)
{
    PSGPU_REQUIRE(d && v, "psgpu_decode_view: NULL argument");
    v->n_utt = d->n_utt; v->total_frames = d->total; v->max_frames = d->max_frames; v->bp_cap = d->bp_cap; v->bss_cap = d->bss_cap;
    v->max_words = d->max_words; v->frame_off = d->frame_off.data();
    v->frame_off_dev = d->d_off; v->feat_dev = d->d_feat; v->topn_cw_dev = d->d_tcw; v->topn_score_dev = d->d_tsc; v->rows_dev = d->d_rows; v->penalties_dev = d->d_pen;
    v->bp_dev = d->d_bp; v->bss_dev = d->d_bss; v->idx_dev = d->d_idx; v->step_dev = d->d_step; v->result_dev = d->d_res;
    v->hyp_dev = d->d_hyp; v->hyp_n_dev = d->d_hn; v->w1_ssid_dev = d->d_w1;
    return PSGPU_OK;
}

int psgpu_decode_search_lag(psgpu_decode_t *d, int32_t lag)
{
    PSGPU_REQUIRE(d && lag >= 0, "psgpu_decode_search_lag: bad argument");
    d->lag_next = lag;
    return PSGPU_OK;
}

int psgpu_decode_table_capacity(psgpu_decode_t *d, int32_t bp_per_frame, int32_t bss_per_frame, int32_t auto_grow)
{
    PSGPU_REQUIRE(d && bp_per_frame >= 0 && bss_per_frame >= 0, "psgpu_decode_table_capacity: bad argument");
    if (bp_per_frame > 0) d->bp_pf = bp_per_frame;
    if (bss_per_frame > 0) d->bss_pf = bss_per_frame;
    d->auto_grow = auto_grow != 0;
    return PSGPU_OK;
}

int32_t psgpu_decode_tables_grown(const psgpu_decode_t *d) { return d ? d->n_grown : 0; }

// (Status 2, the LDS layout's evaluation list: see the first lines of the function.)
// An utterance whose back-pointer table or score stack filled up ended with status 1.  The reference never ends that way:
// it doubles the table (ngram_search.c:449-463, :468-480).  Here: double both allowances, allocate new tables, search the
// call's utterances again on the scores and penalties still in the object's buffers -- until no utterance reports a full
// table, the device has no room for larger ones, or the tables have doubled twelve times.  The larger allowance stays (later calls
// start with it), so a workload pays this once.  Returns PSGPU_OK with `res` holding the final result records.
static int dec_live_mode(psgpu_decode_s *d, bool resume);

static int dec_repeat_with_larger_tables(psgpu_decode_s *d, std::vector<int32_t> &res, hipStream_t st)
{
    const size_t nu = (size_t)d->n_utt, mf = (size_t)d->max_frames;
    // (a live utterance: the repeated search starts at the utterance's first frame again and keeps its state for the next step)
    auto live_again = [&]() { if (d->live) { d->live_mode_next = dec_live_mode(d, false); d->live_ok = d->live_mode_next != 0; d->live_searched += d->live_S; } };
    {   // status 2: the LDS layout's evaluation list (what its pool had left) filled up in some frame.  The slab layout's list holds
        // every channel: the search is switched to it -- for good, this workload needs it -- and the call's search stage repeated.
        bool list_full = false;
        for (size_t u = 0; u < nu && !list_full; ++u) list_full = res[u * 8 + 3] == 2;
        if (list_full && !d->lists) {
            int rc;
            if ((rc = psgpu_fwdtree_use_slab_layout(d->cfg.ft))) return rc;
            ++d->n_grown;
            live_again();
            if ((rc = dec_search(d, d->n_utt, (size_t)d->total, mf, st))) return rc;
            if (d->ev_srch) PSGPU_HIP(hipEventRecord(d->ev_srch, st));
            PSGPU_HIP(hipMemcpyAsync(res.data(), d->d_res, 4 * nu * 8, hipMemcpyDeviceToHost, st));
            PSGPU_HIP(hipStreamSynchronize(st));
        }
    }
    bool no_more_capacity = false;
    for (int pass = 0; pass < 8; ++pass) {               // (a search that goes further with larger tables may meet the other limit)
    for (int round = 0; round < 16 && !no_more_capacity; ++round) {
        // status 4 / 5: a frame listed more tree nodes than the slab layouts' compact channels hold / needed more blocks of the
        // right-context channels' pool than there are: the capacity is doubled (psgpu_fwdtree_grow) -- for good -- and the search repeated
        int32_t grow = 0;
        for (size_t u = 0; u < nu && !grow; ++u) if (res[u * 8 + 3] >= 4 && res[u * 8 + 3] <= 6) grow = res[u * 8 + 3];
        if (!grow) break;
        int rc;
        if ((rc = psgpu_fwdtree_grow(d->cfg.ft, grow))) {
            // nothing left to grow: that status stays as reported (the call succeeds: the error string is not this call's); the
            // other utterances' full tables are still doubled below, once
            psgpu_clear_error();
            no_more_capacity = true;
            break;
        }
        ++d->n_grown;
        live_again();
        if ((rc = dec_search(d, d->n_utt, (size_t)d->total, mf, st))) return rc;
        if (d->ev_srch) PSGPU_HIP(hipEventRecord(d->ev_srch, st));
        PSGPU_HIP(hipMemcpyAsync(res.data(), d->d_res, 4 * nu * 8, hipMemcpyDeviceToHost, st));
        PSGPU_HIP(hipStreamSynchronize(st));
    }
    for (int round = 0; round < 12; ++round) {
        bool full = false;
        for (size_t u = 0; u < nu && !full; ++u) full = res[u * 8 + 3] == 1;
        if (!full) break;
        const size_t cb = 2 * d->cap_bp, cs = 2 * d->cap_bss;
        size_t free_b = 0, total_b = 0;
        if (cb > 0x7ffffff0u / 10 || cs > 0x7ffffff0u || hipMemGetInfo(&free_b, &total_b) != hipSuccess
            || 4 * d->cap_utt * (10 * cb + cs) + ((size_t)256 << 20) > free_b + 4 * d->cap_utt * (10 * d->cap_bp + d->cap_bss))
            return PSGPU_OK;                              // no room: the status stays as reported
        DFREE(d->d_bp); DFREE(d->d_bss);
        d->cap_bp = d->cap_bss = 0;
        int rc;
        if ((rc = dec_alloc((void **)&d->d_bp, 4 * d->cap_utt * 10 * cb)) || (rc = dec_alloc((void **)&d->d_bss, 4 * d->cap_utt * cs))) {
            d->cap_utt = 0;                               // (the next call allocates everything anew)
            return rc;
        }
        d->cap_bp = cb; d->cap_bss = cs;
        if (mf) {                                         // (later calls start with this allowance)
            d->bp_pf = std::max<int32_t>(d->bp_pf, (int32_t)std::min<size_t>((cb + mf - 1) / mf, 1 << 20));
            d->bss_pf = std::max<int32_t>(d->bss_pf, (int32_t)std::min<size_t>((cs + mf - 1) / mf, 1 << 24));
        }
        d->bp_cap = (int32_t)cb; d->bss_cap = (int32_t)cs;
        ++d->n_grown;
        live_again();
        if ((rc = dec_search(d, d->n_utt, (size_t)d->total, mf, st))) return rc;
        if (d->ev_srch) PSGPU_HIP(hipEventRecord(d->ev_srch, st));
        PSGPU_HIP(hipMemcpyAsync(res.data(), d->d_res, 4 * nu * 8, hipMemcpyDeviceToHost, st));
        PSGPU_HIP(hipStreamSynchronize(st));
    }
    bool more = false;
    for (size_t u = 0; u < nu && !more; ++u) more = res[u * 8 + 3] == 1 || (!no_more_capacity && res[u * 8 + 3] >= 4 && res[u * 8 + 3] <= 6);
    if (!more) break;
    }
    return PSGPU_OK;
}

// The host's wait for a batch call's results.  hipStreamSynchronize spins: a host thread waiting 70 ms for the search of 512 x 30 s burns
// a core doing so (bench.py's host.cpu_ms_per_step_per_rank: 1.7 cores busy per GPU in round 4, the fetch's and torch's own
// synchronize) -- with eight ranks a node's cores are not scarce, but a waiting thread has no business running.  A call of
// some size polls an event between 0.1 ms sleeps instead; small calls and the live /
// streams steps, where that lateness is a tenth of the step, keep the spin.  PSGPU_SPIN_WAIT=1: always spin; PSGPU_POLL_WAIT_US: always poll.
static int dec_wait(psgpu_decode_s *d, hipStream_t st)
{
    static const int spin = [] { const char *e = getenv("PSGPU_SPIN_WAIT"); return e ? atoi(e) : 0; }();
    // PSGPU_POLL_WAIT_US=n: EVERY wait -- the live / streams steps' too -- polls between sleeps of n microseconds (a host that serves many
    // decoders per core trades up to n us of a step's latency for the core: 512 streams at 100 ms of audio a step take 0.9 ms a step)
    static const long poll_ns = [] { const char *e = getenv("PSGPU_POLL_WAIT_US"); return e ? 1000L * atol(e) : 0L; }();
    if (spin || (poll_ns <= 0 && (d->live || d->streams || d->total < 100000))) { PSGPU_HIP(hipStreamSynchronize(st)); return PSGPU_OK; }
    // (hipEventSynchronize on an event created with hipEventBlockingSync was measured spinning all the same in this runtime -- the
    //  waiting thread at 100 % of a core, profiles/round5_host_cpu.txt: the event is polled between short sleeps instead, at most
    //  0.1 ms late)
    if (!d->ev_wait) PSGPU_HIP(hipEventCreateWithFlags(&d->ev_wait, hipEventDisableTiming));
    PSGPU_HIP(hipEventRecord(d->ev_wait, st));
    for (;;) {
        const hipError_t q = hipEventQuery(d->ev_wait);
        if (q == hipSuccess) break;
        if (q != hipErrorNotReady) PSGPU_HIP(q);
        const struct timespec ts = { 0, poll_ns > 0 ? (poll_ns < 999999999L ? poll_ns : 999999999L) : 100000L };
        nanosleep(&ts, nullptr);
    }
    return PSGPU_OK;
}

int psgpu_decode_fetch_hyps(psgpu_decode_t *d, int32_t *hyp_n, int32_t *hyp, int32_t *result, void *stream)
{
    PSGPU_REQUIRE(d, "psgpu_decode_fetch_hyps: NULL argument");
    hipStream_t st = (hipStream_t)stream;
    const size_t nu = (size_t)d->n_utt;
    if (nu && d->kind == PSGPU_SCORER_MS && d->searched) {
        // (a frame with fewer than topn densities above WORST_DIST: the reference then keeps stale list ids, ms_gauden.c:438-440,
        //  which the batched kernels do not reproduce -- reported, not hidden)
        int rc = psgpu_ms_batch_check((psgpu_ms_model_t *)d->cfg.scorer, st);
        if (rc != PSGPU_OK) return rc;
    }
    // (streams: a full table ends its stream with status 1 -- the rows of frames already searched are gone, the search cannot be repeated)
    // (the wait comes FIRST: a copy into pageable host memory -- the caller's arrays -- makes the runtime wait for the stream inside
    //  hipMemcpyAsync, spinning)
    if (nu) { const int wrc = dec_wait(d, st); if (wrc != PSGPU_OK) return wrc; }
    if (nu && d->auto_grow && d->searched && !d->pass2 && !d->streams) {
        std::vector<int32_t> res(nu * 8);
        PSGPU_HIP(hipMemcpyAsync(res.data(), d->d_res, 4 * nu * 8, hipMemcpyDeviceToHost, st));
        // (the usual case -- no table was full -- in one wait: the hypotheses travel with the result records)
        if (hyp_n) PSGPU_HIP(hipMemcpyAsync(hyp_n, d->d_hn, 4 * nu * 4, hipMemcpyDeviceToHost, st));
        if (hyp) PSGPU_HIP(hipMemcpyAsync(hyp, d->d_hyp, 4 * nu * d->max_words * 4, hipMemcpyDeviceToHost, st));
        PSGPU_HIP(hipStreamSynchronize(st));             // (the copies alone: the stream's work is over)
        bool again = false;
        for (size_t u = 0; u < nu && !again; ++u) again = res[u * 8 + 3] != 0;
        if (!again) {
            if (result) memcpy(result, res.data(), 4 * nu * 8);
            return PSGPU_OK;
        }
        int rc = dec_repeat_with_larger_tables(d, res, st);
        if (rc != PSGPU_OK) return rc;
        if (result) memcpy(result, res.data(), 4 * nu * 8);
        result = nullptr;
    }
    if (nu) {
        if (hyp_n) PSGPU_HIP(hipMemcpyAsync(hyp_n, d->d_hn, 4 * nu * 4, hipMemcpyDeviceToHost, st));
        if (hyp) PSGPU_HIP(hipMemcpyAsync(hyp, d->d_hyp, 4 * nu * d->max_words * 4, hipMemcpyDeviceToHost, st));
        if (result) PSGPU_HIP(hipMemcpyAsync(result, d->pass2 ? d->d_res2 : d->d_res, 4 * nu * 8, hipMemcpyDeviceToHost, st));
    }
    PSGPU_HIP(hipStreamSynchronize(st));
    return PSGPU_OK;
}

int psgpu_decode_second_pass(psgpu_decode_t *d, psgpu_fwdflat_t *ff, void *stream)
{
    PSGPU_REQUIRE(d && ff, "psgpu_decode_second_pass: NULL argument");
    PSGPU_REQUIRE(d->first_called, "psgpu_decode_second_pass: no first pass in this object (psgpu_decode_first_pass* comes first)");
    // a call of nothing but empty (or too short) utterances, or of none: the first pass launched no search and left empty result
    // records; the second pass of nothing is nothing -- those records stand (as the first-pass-only path returns them)
    if (d->n_utt == 0 || d->total == 0) { d->pass2 = false; return PSGPU_OK; }
    PSGPU_REQUIRE(d->searched, "psgpu_decode_second_pass: the first pass of this call did not complete");
    PSGPU_REQUIRE(d->kind == PSGPU_SCORER_PTM, "psgpu_decode_second_pass: the device second pass scores from the PTM scorer's lists");
    PSGPU_REQUIRE(!d->compall, "psgpu_decode_second_pass: the device second pass normalises over its own senone lists (-compallsen no)");
    PSGPU_REQUIRE(psgpu_ptm_model_view(d->cfg.model, &d->view) == PSGPU_OK, "psgpu_decode_second_pass: no view of the PTM model");
    PSGPU_REQUIRE(d->last_lag == 0, "psgpu_decode_second_pass: the first pass stopped short of the utterances' ends (psgpu_decode_search_lag)");
    // (the second pass re-reads the call's feature rows and frame offsets; a front end run ahead into them has replaced them with the
    //  NEXT call's -- psgpu_decode_front_end_ahead comes after this call's second pass, which returns with its search finished)
    PSGPU_REQUIRE(!d->fe_ahead, "psgpu_decode_second_pass: the next call's front end has been run ahead into this object's feature rows "
                  "(psgpu_decode_front_end_ahead comes after the second pass)");
    hipStream_t st = (hipStream_t)stream;
    const size_t nu = (size_t)d->n_utt, mf = (size_t)d->max_frames;
    d->pass2 = false;
    int rc;
    {   // the first pass's tables complete (a full table: larger ones and the search again, as psgpu_decode_fetch_hyps would)
        std::vector<int32_t> res(nu * 8);
        PSGPU_HIP(hipMemcpyAsync(res.data(), d->d_res, 4 * nu * 8, hipMemcpyDeviceToHost, st));
        PSGPU_HIP(hipStreamSynchronize(st));
        if (d->auto_grow && (rc = dec_repeat_with_larger_tables(d, res, st))) return rc;
        for (size_t u = 0; u < nu; ++u)
            PSGPU_REQUIRE(res[u * 8 + 3] == 0, "psgpu_decode_second_pass: utterance %zu's first pass ended with status %d", u, res[u * 8 + 3]);
    }
    size_t cb = d->cap_bp, cs = d->cap_bss;              // the second pass's tables start at the first pass's capacities
    for (int round = 0;; ++round) {
        if (nu > d->cap2_utt || cb > d->cap2_bp || cs > d->cap2_bss || mf > d->cap2_mf) {
            DFREE(d->d_bp2); DFREE(d->d_bss2); DFREE(d->d_idx2); DFREE(d->d_step2); DFREE(d->d_res2); DFREE(d->d_seed2);
            d->cap2_utt = d->cap2_bp = d->cap2_bss = d->cap2_mf = 0;
            const size_t cu = std::max(nu, d->cap_utt), cm = std::max(mf, d->cap_mf);
            if ((rc = dec_alloc((void **)&d->d_bp2, 4 * cu * 10 * cb)) || (rc = dec_alloc((void **)&d->d_bss2, 4 * cu * cs))
                || (rc = dec_alloc((void **)&d->d_idx2, 4 * cu * (cm + 2))) || (rc = dec_alloc((void **)&d->d_step2, 4 * cu * std::max<size_t>(cm, 1) * 4))
                || (rc = dec_alloc((void **)&d->d_res2, 4 * cu * 8))
                || (rc = dec_alloc((void **)&d->d_seed2, 4 * cu * (size_t)d->n_chain * d->topn)))
                return rc;
            d->cap2_utt = cu; d->cap2_bp = cb; d->cap2_bss = cs; d->cap2_mf = cm;
        }
        d->bp_cap2 = (int32_t)d->cap2_bp; d->bss_cap2 = (int32_t)d->cap2_bss;
        hipLaunchKernelGGL(dec_pass2_seed_kernel, dim3((unsigned)nu), dim3(128), 0, st, d->d_tcw, d->d_off, d->total, d->n_chain, d->topn,
                           d->cfg.pl_window + 2, d->d_seed2);
        PSGPU_HIP(hipGetLastError());
        const uint8_t *open_flags = nullptr;
        if ((rc = psgpu_ptm_batch_open_flags(d->cfg.model, st, &open_flags))) return rc;
        if ((rc = psgpu_fwdflat_search_feats_lists_dev(ff, &d->view, d->d_feat, d->d_seed2, d->d_tsc, d->d_tcw, open_flags, d->total, d->d_off,
                                                       d->n_utt, d->max_frames, d->bp_cap, d->d_bp, d->d_res, d->d_w1, d->bp_cap2, d->bss_cap2,
                                                       d->d_bp2, d->d_bss2, d->d_idx2, d->d_step2, d->d_res2, st)))
            return rc;
        std::vector<int32_t> res2(nu * 8);
        PSGPU_HIP(hipMemcpyAsync(res2.data(), d->d_res2, 4 * nu * 8, hipMemcpyDeviceToHost, st));
        PSGPU_HIP(hipStreamSynchronize(st));
        bool full = false;
        for (size_t u = 0; u < nu && !full; ++u) full = res2[u * 8 + 3] == 1;
        if (!full || !d->auto_grow || round >= 12) break;
        cb = 2 * d->cap2_bp; cs = 2 * d->cap2_bss;          // (the reference grows these tables on demand as well)
        size_t free_b = 0, total_b = 0;
        if (cb > 0x7ffffff0u / 10 || cs > 0x7ffffff0u || hipMemGetInfo(&free_b, &total_b) != hipSuccess
            || 4 * d->cap2_utt * (10 * cb + cs) + ((size_t)256 << 20) > free_b + 4 * d->cap2_utt * (10 * d->cap2_bp + d->cap2_bss))
            break;
        ++d->n_grown;
    }
    // the hypotheses of this pass in the place of the first's (ngram_search_find_exit + backtrace on the second pass's table)
    if ((rc = psgpu_fwdtree_backtrace_dev(d->cfg.ft, d->d_bp2, d->d_idx2, d->d_res2, d->n_utt, d->max_frames, d->bp_cap2, d->max_words,
                                          d->d_hyp, d->d_hn, st)))
        return rc;
    d->pass2 = true;
    return PSGPU_OK;
}

int psgpu_decode_fetch_tables(psgpu_decode_t *d, int32_t u, int32_t n_bp, int32_t n_bss, int32_t n_idx, int32_t *bp, int32_t *bss,
                              int32_t *idx, void *stream)
{
    return psgpu_decode_fetch_tables_range(d, u, 0, n_bp, 0, n_bss, 0, n_idx, bp, bss, idx, stream);
}

// entries [bp0, bp0 + n_bp) of the table (ten columns, n_bp apart on the host), [bss0, bss0 + n_bss) of the score stack,
// [idx0, idx0 + n_idx) of the frame marks: the tables only grow while an utterance is in progress (psgpu_decode_live_step), a caller
// that holds what an earlier read-out returned asks for the rest
int psgpu_decode_fetch_tables_range(psgpu_decode_t *d, int32_t u, int32_t bp0, int32_t n_bp, int32_t bss0, int32_t n_bss, int32_t idx0,
                                    int32_t n_idx, int32_t *bp, int32_t *bss, int32_t *idx, void *stream)
{
    PSGPU_REQUIRE(d && u >= 0 && u < d->n_utt && bp0 >= 0 && bss0 >= 0 && idx0 >= 0, "psgpu_decode_fetch_tables: bad argument");
    // (after psgpu_decode_second_pass: that pass's tables, as psgpu_decode_fetch_hyps returns its hypotheses and result records)
    const int32_t bcap = d->pass2 ? d->bp_cap2 : d->bp_cap, scap = d->pass2 ? d->bss_cap2 : d->bss_cap;
    const int32_t *const t_bp = d->pass2 ? d->d_bp2 : d->d_bp, *const t_bss = d->pass2 ? d->d_bss2 : d->d_bss, *const t_idx = d->pass2 ? d->d_idx2 : d->d_idx;
    PSGPU_REQUIRE(n_bp >= 0 && bp0 + n_bp <= bcap && n_bss >= 0 && bss0 + n_bss <= scap && n_idx >= 0 && idx0 + n_idx <= d->max_frames + 2,
                  "psgpu_decode_fetch_tables: bad argument");
    hipStream_t st = (hipStream_t)stream;
    if (bp && n_bp)       // ten columns, bp_cap apart on the device, n_bp apart on the host
        PSGPU_HIP(hipMemcpy2DAsync(bp, 4 * (size_t)n_bp, t_bp + (size_t)u * 10 * bcap + bp0, 4 * (size_t)bcap, 4 * (size_t)n_bp, 10,
                                   hipMemcpyDeviceToHost, st));
    if (bss && n_bss) PSGPU_HIP(hipMemcpyAsync(bss, t_bss + (size_t)u * scap + bss0, 4 * (size_t)n_bss, hipMemcpyDeviceToHost, st));
    if (idx && n_idx) PSGPU_HIP(hipMemcpyAsync(idx, t_idx + (size_t)u * (d->max_frames + 2) + idx0, 4 * (size_t)n_idx, hipMemcpyDeviceToHost, st));
    PSGPU_HIP(hipStreamSynchronize(st));
    return PSGPU_OK;
}

}  // extern "C"

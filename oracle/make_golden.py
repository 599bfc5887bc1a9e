#!/usr/bin/env python3
"""Generate the committed golden fixtures under tests/golden/ from the
UNMODIFIED reference compiled into oracle/_ref (run `make -C oracle` first).

TEST INFRASTRUCTURE.  Runs only in the build container (needs oracle/_ref and
therefore /root/reference); the fixtures it writes are what travels.

Fixtures:
  en_us_ptm_tables.npz     model tables as the reference holds them after init
  ptm_<case>.npz           PTM scorer goldens (compallsen) for several inputs
  senlog_default.npz       every frame_eval call of a default-mode decode
  decode_<mode>.npz        hypothesis + segmentation goldens
  hmm_<model>.npz          hmm_vit_eval state before/after each step (3- and 5-state, mpx and not)

Large per-frame outputs are stored as 64-bit FNV-1a row hashes for every frame
plus the full rows of a frame sample (tests recompute the hashes).
"""
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from psgb import read_psgb  # noqa: E402
from pso import row_hash  # noqa: E402

REF = os.path.join(HERE, "_ref")
GOLD = os.path.join(ROOT, "tests", "golden")
MODEL = os.path.join(REF, "model", "en-us")
LM = os.path.join(REF, "data", "turtle.lm.bin")
DIC = os.path.join(REF, "data", "turtle.dic")
RAW = os.path.join(REF, "data", "goforward.raw")
SEED = 20260921


def ref_dump(cmd, *args, extra=(), model=None, lm=None, dic=None):
    with tempfile.NamedTemporaryFile(suffix=".psgb", delete=False) as fh:
        out = fh.name
    argv = [os.path.join(REF, "ref_dump"), cmd, out, model or MODEL, lm or LM, dic or DIC] + [str(a) for a in args]
    if extra:
        argv += ["--"] + [str(e) for e in extra]
    subprocess.check_call(argv)
    d = read_psgb(out)
    os.unlink(out)
    return d


def synth_feats(t, n, rng):
    """x[t][d] ~ N(mu_d, sigma_d^2) with mu/sigma the
    per-dimension mean/std of the model means."""
    n_mgau, n_feat, n_den = int(t["n_mgau"][0]), int(t["n_feat"][0]), int(t["n_density"][0])
    fl = int(t["featlen"][0])
    mean = t["mean"].reshape(n_mgau, n_feat, n_den, fl)
    mu = mean.mean(axis=(0, 2)).reshape(-1)
    sd = mean.std(axis=(0, 2)).reshape(-1)
    return (mu + sd * rng.standard_normal((n, mu.size))).astype(np.float32)


def adversarial_feats(t, n, rng):
    """(C): every stream sits on (or a hair off) a codeword mean of a random
    codebook, so distances land on/near det values and near-ties appear."""
    n_mgau, n_feat, n_den = int(t["n_mgau"][0]), int(t["n_feat"][0]), int(t["n_density"][0])
    fl = int(t["featlen"][0])
    mean = t["mean"].reshape(n_mgau, n_feat, n_den, fl)
    x = np.empty((n, n_feat, fl), np.float32)
    for i in range(n):
        for f in range(n_feat):
            cb = rng.integers(n_mgau); cw = rng.integers(n_den)
            eps = (0.0 if i % 3 == 0 else 1e-3 * rng.standard_normal(fl))
            x[i, f] = mean[cb, f, cw] + eps
    return x.reshape(n, n_feat * fl)


def ptm_case(name, feats, seglen, carry, dup=0, full_topn=False, sample=16, model=None, more=None):
    with tempfile.NamedTemporaryFile(suffix=".f32", delete=False) as fh:
        feats.astype(np.float32).tofile(fh)
        fpath = fh.name
    d = ref_dump("ptm", fpath, seglen, carry, dup, model=model)
    os.unlink(fpath)
    T = feats.shape[0]
    idx = np.unique(np.linspace(0, T - 1, sample).astype(np.int64))
    topn = np.concatenate([d["topn_cw"].reshape(T, -1).astype(np.int32),
                           d["topn_raw"].reshape(T, -1)], axis=1)
    out = dict(feat=feats.astype(np.float32), seglen=np.int32(seglen), carry=np.int32(carry),
               dup=np.int32(dup),
               senscr_hash=row_hash(d["senscr"]), topn_hash=row_hash(topn),
               sample_idx=idx, senscr_sample=d["senscr"][idx],
               topn_cw_sample=d["topn_cw"][idx], topn_raw_sample=d["topn_raw"][idx],
               topn_norm_sample=d["topn_norm"][idx])
    if full_topn:
        out["topn_cw"] = d["topn_cw"]; out["topn_raw"] = d["topn_raw"]
    if more:
        out.update(more)
    np.savez_compressed(os.path.join(GOLD, "ptm_%s.npz" % name), **out)
    print("ptm_%s: T=%d" % (name, T))
    return d


def write_clustered_sendump(path, mixw8, n_clust=16):
    """A sendump with 4-bit clustered mixture weights in the layout read_sendump (ptm_mgau.c:457-654) expects: sixteen
    cluster values, then per (stream, density) a row of (n_sen + 1) / 2 bytes, the even senone's cluster index in the low
    nibble.  The clusters are sixteen quantiles of the 8-bit weights."""
    import struct
    n_feat, n_den, n_sen = mixw8.shape
    cb = np.unique(np.quantile(mixw8.ravel(), np.linspace(0, 1, n_clust)).astype(np.int64))
    while cb.size < n_clust:                        # (ties between quantiles: fill with unused values)
        cb = np.unique(np.concatenate([cb, [int(cb.max()) + 1 if cb.max() < 255 else int(cb.min()) - 1]]))
    cb = np.sort(cb)[:n_clust].astype(np.uint8)
    idx = np.abs(mixw8[..., None].astype(np.int32) - cb.astype(np.int32)).argmin(axis=-1).astype(np.uint8)
    if n_sen & 1:
        idx = np.concatenate([idx, np.zeros((n_feat, n_den, 1), np.uint8)], axis=2)
    packed = (idx[..., 0::2] | (idx[..., 1::2] << 4)).astype(np.uint8)

    def lstr(txt):
        b = txt.encode() + b"\0"
        return struct.pack("<i", len(b)) + b
    with open(path, "wb") as fh:
        fh.write(lstr("V6 Senone Probs, Smoothed, Normalized"))
        fh.write(lstr("(HMM file format)"))
        for h in ("feature_count %d" % n_feat, "mixture_count %d" % n_den, "model_count %d" % n_sen,
                  "cluster_count %d" % n_clust, "cluster_bits 4", "logbase 1.0001", "mixw_shift 10"):
            fh.write(lstr(h))
        fh.write(struct.pack("<i", 0))
        fh.write(cb.tobytes())
        fh.write(packed.tobytes())
    return cb, packed


def ptm_4bit():
    """PTM scoring from a 4-bit clustered sendump (ptm_mgau.c:375-379, with its nibble selection by the low bit of the
    BYTE): the en-us model with its sendump re-quantised.  The golden carries the packed weights and the cluster values."""
    t = np.load(os.path.join(GOLD, "en_us_ptm_tables.npz"))
    gof = ref_dump("feats", RAW)["feat"]
    d = tempfile.mkdtemp()
    for f in os.listdir(MODEL):
        if f != "sendump":
            os.symlink(os.path.join(MODEL, f), os.path.join(d, f))
    cb, packed = write_clustered_sendump(os.path.join(d, "sendump"), t["mixw"])
    tt = ref_dump("tables", model=d)
    assert int(tt["mixw_is_4bit"][0]) == 1 and np.array_equal(tt["mixw"], packed) and np.array_equal(tt["mixw_cb"], cb)
    ptm_case("4bit_goforward", gof, gof.shape[0], 0, sample=24, model=d, more=dict(mixw4=packed, mixw_cb=cb))


def senlog_case(name, nrep, extra=(), inp=None, **kw):
    d = ref_dump("senlog", inp or RAW, nrep, extra=extra, **kw)
    n = int(d["n_calls"][0])
    idx = np.unique(np.linspace(0, n - 1, 24).astype(np.int64))
    out = {k: v for k, v in d.items() if k.startswith("utt") or k.startswith("call_")}
    scr = out.pop("call_scr")
    out["call_scr_hash"] = row_hash(scr)
    out["sample_idx"] = idx
    out["call_scr_sample"] = scr[idx]
    out["extra"] = np.array(list(extra), dtype="U")
    np.savez_compressed(os.path.join(GOLD, "senlog_%s.npz" % name), **out)
    print("senlog_%s: %d calls, hyp=%r" % (name, n, bytes(d["utt0_hyp"]).decode()))


def decode_case(name, extra=()):
    d = ref_dump("decode", RAW, extra=extra)
    d["extra"] = np.array(list(extra), dtype="U")
    np.savez_compressed(os.path.join(GOLD, "decode_%s.npz" % name), **d)
    print("decode_%s: hyp=%r score=%d" % (name, bytes(d["hyp"]).decode(), int(d["hyp_score"][0])))


def hmm_case(name, model, lm, dic, n_hmm, n_steps, seed):
    with tempfile.NamedTemporaryFile(suffix=".psgb", delete=False) as fh:
        out = fh.name
    subprocess.check_call([os.path.join(REF, "ref_dump"), "hmm", out, model, lm, dic,
                           str(n_hmm), str(n_steps), str(seed)])
    d = read_psgb(out)
    os.unlink(out)
    np.savez_compressed(os.path.join(GOLD, "hmm_%s.npz" % name), **d)
    print("hmm_%s: n_emit=%d n_hmm=%d steps=%d" % (name, int(d["n_emit"][0]), n_hmm, n_steps))


def main():
    os.makedirs(GOLD, exist_ok=True)
    t = ref_dump("tables")
    np.savez_compressed(os.path.join(GOLD, "en_us_ptm_tables.npz"), **t)
    print("tables:", {k: v.shape for k, v in t.items() if v.size > 1})

    gof = ref_dump("feats", RAW)["feat"]
    rng = np.random.default_rng(SEED)
    ptm_case("goforward", gof, gof.shape[0], 0, full_topn=True)
    ptm_case("goforward_x2_carry", np.concatenate([gof, gof]), gof.shape[0], 1)
    ptm_case("synth", synth_feats(t, 1500, rng), 1500, 0)
    ptm_case("synth_utts", synth_feats(t, 6 * 100, rng), 100, 0)
    ptm_case("adversarial", adversarial_feats(t, 512, rng), 512, 0)
    ptm_case("dup_ties", adversarial_feats(t, 384, rng), 128, 1, dup=1)

    senlog_case("default", 2)
    senlog_case("fwdtree_only", 1, extra=("fwdflat", "no", "bestpath", "no"))
    ptm_topn_only()
    decode_case("default")
    decode_case("fwdtree_only", extra=("fwdflat", "no", "bestpath", "no"))
    decode_case("compallsen_plw0", extra=("compallsen", "yes", "pl_window", "0"))
    hmm_only()


TD_MODEL = os.path.join(REF, "model", "tidigits")
TD_DATA = os.path.join(REF, "data", "tidigits")
TD = dict(model=TD_MODEL, lm=os.path.join(TD_DATA, "tidigits.lm.bin"), dic=os.path.join(TD_DATA, "tidigits.dic"))


def semi_only():
    """s2_semi scorer (tidigits: 4 streams x 256 densities, 4-bit clustered mixw,
    5-state HMMs): tables + every frame_eval call of real decodes."""
    t = ref_dump("tables_semi", **TD)
    np.savez_compressed(os.path.join(GOLD, "semi_tidigits_tables.npz"), **t)
    print("semi tables:", {k: v.shape for k, v in t.items() if v.size > 1})
    a = os.path.join(TD_DATA, "man.ah.111a.mfc")
    b = os.path.join(TD_DATA, "woman.ak.276317oa.mfc")
    senlog_case("tidigits_default", 2, inp=a, **TD)
    senlog_case("tidigits_beam", 1, inp=b, extra=("topn_beam", "20,35,10,20"), **TD)
    senlog_case("tidigits_topn6_ds2", 1, inp=a, extra=("topn", "6", "ds", "2"), **TD)
    senlog_case("tidigits_topn7_call", 1, inp=a, extra=("topn", "7", "compallsen", "yes"), **TD)
    senlog_case("tidigits_topn2", 1, inp=b, extra=("topn", "2", "pl_window", "0"), **TD)
    senlog_case("tidigits_topn8_list", 1, inp=a, extra=("topn", "8", "fwdflat", "no"), **TD)   # 4-bit `_any` kernel, active lists


def stage_en_us_ms():
    from stage_ms_model import stage
    return stage(MODEL, os.path.join(GOLD, "en_us_ptm_tables.npz"), os.path.join(REF, "model", "en-us-ms"))


def ms_only():
    """ms scorer: an4_ci_cont (102 codebooks x 1 density x 39 dims, compute_dist_all
    path) and en-us-ms (42 codebooks x 3 streams x 128 densities, top-N scan path)."""
    an4 = os.path.join(REF, "model", "an4_ci_cont")
    t = ref_dump("tables_ms", model=an4)
    np.savez_compressed(os.path.join(GOLD, "ms_an4_tables.npz"), **t)
    print("ms an4 tables:", {k: v.shape for k, v in t.items() if v.size > 1})
    senlog_case("ms_an4_default", 1, model=an4)
    senlog_case("ms_an4_compall_aw2", 1, model=an4, extra=("compallsen", "yes", "aw", "2"))
    ems = stage_en_us_ms()
    x = ("senmgau", ".ptm.")
    t = ref_dump("tables_ms", model=ems, extra=x)
    np.savez_compressed(os.path.join(GOLD, "ms_en_us_tables.npz"), **t)
    print("ms en-us tables:", {k: v.shape for k, v in t.items() if v.size > 1})
    senlog_case("ms_en_us_default", 1, model=ems, extra=x)
    senlog_case("ms_en_us_topn2_call", 1, model=ems, extra=x + ("topn", "2", "compallsen", "yes", "fwdflat", "no"))


def ptm_topn_only():
    # PTM with other top-N sizes (the any-shape per-call path)
    senlog_case("ptm_topn2", 1, extra=("topn", "2", "fwdflat", "no"))
    senlog_case("ptm_topn6_ds2", 1, extra=("topn", "6", "ds", "2", "fwdflat", "no"))


def dynfeat_only():
    d = ref_dump("dynfeat", os.path.join(REF, "data", "goforward.mfc"))
    np.savez_compressed(os.path.join(GOLD, "dynfeat_goforward.npz"), **d)
    print("dynfeat_goforward:", d["cep"].shape, "->", d["feat"].shape)


def mfcc_only():
    """Front-end goldens: the reference's tables + cepstra.  `mfcc` uses the en-us decoder's
    front end; the variants a stand-alone fe_t (ref_dump mfcc_cfg) configured by key/value."""
    def clip(src, n0, n1):
        pcm = np.fromfile(os.path.join(REF, "data", src), dtype=np.int16)[n0:n1]
        fh = tempfile.NamedTemporaryFile(suffix=".raw", delete=False)
        pcm.tofile(fh); fh.close()
        return fh.name

    d = ref_dump("mfcc", os.path.join(REF, "data", "goforward.raw"), 2)
    np.savez_compressed(os.path.join(GOLD, "mfcc_en_us_goforward.npz"), **d)
    print("mfcc_en_us_goforward:", d["cep"].shape)
    en = ("lowerf", "130", "upperf", "6800", "nfilt", "25", "transform", "dct", "lifter", "22", "remove_noise", "yes")
    cases = {
        "legacy_dc": (("remove_dc", "yes"), (0, 12000)),                     # built-in defaults: 40 filters, legacy transform
        "htk_40": (("transform", "htk", "lifter", "22", "remove_noise", "yes"), (3000, 14000)),
        "logspec": (en + ("nfilt", "31", "logspec", "yes"), (0, 9000)),
        "smoothspec": (en + ("smoothspec", "yes"), (0, 9000)),
        "nfft1024": (en + ("nfft", "1024", "wlen", "0.032", "frate", "125", "alpha", "0"), (1000, 13000)),
        "short": (en, (5000, 5300)),                    # fewer samples than one frame: only the fe_end_utt frame
        "exact": (en, (5000, 5000 + 410 + 160 * 7)),    # no left-over samples beyond the last full frame
    }
    for name, (extra, (n0, n1)) in cases.items():
        raw = clip("numbers.raw", n0, n1)
        d = ref_dump("mfcc_cfg", raw, 2, lm="-", dic="-", extra=extra)
        os.unlink(raw)
        np.savez_compressed(os.path.join(GOLD, "mfcc_%s.npz" % name), **d)
        print("mfcc_%s:" % name, [int(v) for v in d["par"][:12]], d["cep"].shape)


FT_STATIC = ["node_ci", "node_ci2", "node_ssid", "node_tmat", "node_child", "node_sib", "node_penult_wid",
             "homophone_set", "w1_wid", "w1_ci", "w1_ci2", "w1_ssid", "w1_tmat", "w1_mpx", "dict_pronlen", "dict_first",
             "dict_last", "dict_last2", "dict_basewid", "dict_filler", "dict_real", "rssid_n", "rssid_ssid", "rssid_cimap",
             "ldiph_lc", "tp", "sseq", "ci_tmat", "lm"]


def fwdtree_only():
    """Lexicon-tree search goldens (ref_dump fwdtree): the static search tables of a (model, dictionary,
    LM) triple once, and per decode the trace the search consumed (senone scores, phone-loop
    penalties) with the back-pointer table it produced."""
    tdm = os.path.join(REF, "model", "tidigits")
    tdl = os.path.join(REF, "data", "tidigits")
    base = ("fwdflat", "no", "bestpath", "no")
    cases = [
        ("en_us_turtle", "goforward", dict(), "goforward.raw", base),
        ("en_us_turtle", "numbers", dict(), "numbers.raw", base),
        ("en_us_turtle", "goforward_maxhmmpf60_maxwpf3", dict(), "goforward.raw", base + ("maxhmmpf", "60", "maxwpf", "3")),
        ("en_us_turtle", "something_plwindow0", dict(), "something.raw", base + ("pl_window", "0")),
        ("tidigits", "man_ah_2934za", dict(model=tdm, lm=os.path.join(tdl, "tidigits.lm.bin"), dic=os.path.join(tdl, "tidigits.dic")),
         os.path.join("tidigits", "man.ah.2934za.mfc"), base),
    ]
    done = set()
    for static, name, kw, audio, extra in cases:
        d = ref_dump("fwdtree", os.path.join(REF, "data", audio), extra=extra, **kw)
        if static not in done:
            np.savez_compressed(os.path.join(GOLD, "fwdtree_static_%s.npz" % static), **{k: d[k] for k in FT_STATIC})
            done.add(static)
        tr = {k: v for k, v in d.items() if k not in FT_STATIC}
        tr["static"] = np.frombuffer(static.encode(), np.uint8)
        np.savez_compressed(os.path.join(GOLD, "fwdtree_trace_%s.npz" % name), **tr)
        print("fwdtree", name, "steps", int(d["n_steps"][0]), "bp", d["bp"].shape[0])



def fwdtree_topn_only():
    """The first pass with the scorer's other knobs (-topn 2; -topn 6 -ds 2: ptm_mgau.c:804-896, config_macro.h:384): what the search
    PRODUCED -- back-pointer table, score stack, frame marks, hypothesis -- without the trace it consumed (the inputs are
    goforward.raw and the knobs): the device pipeline decodes the same PCM with a scorer of that shape."""
    base = ("fwdflat", "no", "bestpath", "no")
    keep = ["par", "hyp", "hyp_score", "seg", "seg_words", "bp", "bscore_stack", "bp_table_idx", "n_frame", "step_best", "step_lpbest", "step_bpidx"]
    for name, extra in (("goforward_topn2", ("topn", "2")), ("goforward_topn6_ds2", ("topn", "6", "ds", "2")), ("numbers_topn1", ("topn", "1")),
                        ("numbers_topn8_ds3", ("topn", "8", "ds", "3"))):
        d = ref_dump("fwdtree", os.path.join(REF, "data", name.split("_")[0] + ".raw"), extra=base + extra)
        tr = {k: d[k] for k in keep}
        tr["knobs"] = np.array(extra)
        np.savez_compressed(os.path.join(GOLD, "fwdtree_result_%s.npz" % name), **tr)
        print("fwdtree result", name, "bp", d["bp"].shape[0], "hyp", bytes(d["hyp"]).decode(), int(d["hyp_score"][0]),
              os.path.getsize(os.path.join(GOLD, "fwdtree_result_%s.npz" % name)))


def _arpa_ngrams(path, limit):
    """word tuples of the 2- and 3-gram sections of an ARPA file (plain or .gz)"""
    import gzip
    op = gzip.open if path.endswith(".gz") else open
    out, sec = [], 0
    with op(path, "rt", errors="replace") as fh:
        for ln in fh:
            ln = ln.strip()
            if ln.startswith("\\") and ln.endswith("-grams:"):
                sec = int(ln[1:ln.index("-")]); continue
            if sec in (2, 3) and ln and not ln.startswith("\\"):
                f = ln.split()
                if len(f) >= 1 + sec:
                    out.append(tuple(f[1:1 + sec]))
    rng = np.random.default_rng(5)
    if len(out) > limit:
        out = [out[i] for i in sorted(rng.choice(len(out), limit, replace=False))]
    return out


def lm_set_only():
    """A model SET from the reference's own -lmctl fixture (test/unit/test_ngram/100.lmctl: three models, two word classes defined in
    100.probdef), looked up with one member selected and interpolated (ref_dump lm_set): the members' tables, the set's weights and
    log-add table, queries -- n-grams the members hold + random triples -- and ngram_tg_score's answers."""
    ngdir = "/root/reference/test/unit/test_ngram"
    lmctl = os.path.join(ngdir, "100.lmctl")
    d0 = ref_dump("lm_set", lmctl, "interp", 0, 6.5, 0.65, model="-", lm="-", dic="-")
    words = bytes(d0["words"]).decode().split("\n")[:-1]
    wid = {w: i for i, w in enumerate(words)}
    grams = []
    for f in ("turtle.lm", "100.lm.gz", "102.lm.gz"):
        grams += _arpa_ngrams(os.path.join(ngdir, f), 700)
    qs = []
    rng = np.random.default_rng(9)
    for g_ in grams:
        ids = [wid.get(w, wid.get(w.lower(), -2)) for w in g_]
        if min(ids) < 0:
            continue
        qs.append((ids[-1], ids[-2], ids[-3] if len(ids) > 2 else -1))
    cls = [i for i, w in enumerate(words) if ":" in w]              # the class words: as the word looked up and as history
    for c in cls:
        for _ in range(12):
            a, b = int(rng.integers(0, len(words))), int(rng.integers(0, len(words)))
            qs += [(c, a, b), (a, c, b), (a, b, c), (c, -1, -1), (a, c, -1)]
    for _ in range(600):
        a, b, c = (int(v) for v in rng.integers(0, len(words), 3))
        qs.append((a, b if rng.random() > 0.15 else -1, c if rng.random() > 0.3 else -1))
    qs = [(a, b, (c if b >= 0 else -1)) for a, b, c in qs]
    q = np.array(qs, np.int32)
    with tempfile.NamedTemporaryFile(suffix=".q", delete=False) as fh:
        q.tofile(fh); qpath = fh.name
    out = {}
    for mi, mode in enumerate(("select:100", "select:102", "select:turtle", "interp", "interp:0.5,0.3,0.2")):
        d = ref_dump("lm_set", lmctl, mode, q.shape[0], 6.5, 0.65, qpath, model="-", lm="-", dic="-")
        assert np.array_equal(d["queries"], q)
        if mi == 0:
            for k, v in d.items():
                if k.startswith("m") and k[1].isdigit() or k in ("n_models", "n_words", "addtab", "add_zero", "set_log_zero", "words", "queries"):
                    out[k] = v
        out["mode%d" % mi] = np.frombuffer(mode.encode(), np.uint8)
        out["cur%d" % mi] = d["cur"]; out["lweights%d" % mi] = d["lweights"]; out["scores%d" % mi] = d["scores"]; out["n_used%d" % mi] = d["n_used"]
        print("lm_set", mode, "queries", q.shape[0], "n_used", np.unique(d["n_used"], return_counts=True), "log_zero answers", int((d["scores"] <= d["set_log_zero"][0]).sum()))
    os.unlink(qpath)
    np.savez_compressed(os.path.join(GOLD, "lm_set_100.npz"), **out)
    print(os.path.getsize(os.path.join(GOLD, "lm_set_100.npz")))


FEAT_CASES = [("s2_4x", "batch", 0, "none", 0, "-"), ("s3_1x39", "batch", 1, "max", 0, "-"), ("1s_c_d_dd", "batch", 0, "none", 29, "0-9/10-19/20-28"),
              ("1s_c_d_ld_dd", "none", 0, "none", 0, "-"), ("1s_c_d", "batch", 1, "none", 0, "-"), ("1s_c", "none", 0, "max", 0, "-"),
              ("1s_3c", "batch", 0, "none", 40, "-"), ("5,8:2", "batch", 0, "none", 0, "-"), ("1s_c_d_dd", "batch", 0, "none", 0, "0-12/13-25/26-38"),
              ("1s_4c", "none", 0, "none", 0, "-"), ("1s_12c_12d_3p_12dd", "batch", 0, "max", 20, "3,1,0,19/5-8")]


def feat_types_only():
    """feat_s2mfc2feat_live(begin = end = TRUE) for every feature type feat_init knows, with batch CMN / unit variance / agc max, a
    linear transform and subvector specifications (ref_dump dynfeat_cfg): the cepstra once, per case the settings, the transform
    and the feature frames."""
    out = {}
    for ci, (typ, cmn, vn, agc, ldadim, sv) in enumerate(FEAT_CASES):
        d = ref_dump("dynfeat_cfg", os.path.join(REF, "data", "goforward.mfc"), typ, cmn, vn, agc, ldadim, sv, model="-", lm="-", dic="-")
        k = "c%d_" % ci
        out["cep"] = d["cep"]
        out[k + "feat"] = d["feat"]
        out[k + "cfg"] = np.array([typ, cmn, str(vn), agc, str(ldadim), sv])
        if "lda" in d:
            out[k + "lda"] = d["lda"]
        if "subvec" in d:
            out[k + "subvec"] = d["subvec"]
        print("feat", typ, cmn, vn, agc, ldadim, sv, d["feat"].shape, int(d["window"][0]))
    out["n_cases"] = np.array([len(FEAT_CASES)], np.int32)
    np.savez_compressed(os.path.join(GOLD, "feat_types.npz"), **out)
    print(os.path.getsize(os.path.join(GOLD, "feat_types.npz")))


LIVE_CASES = [("goforward", 2, [1600]), ("goforward", 3, [317]), ("numbers", 3, [800, 2400]), ("goforward", 2, [100, 5000, 333, 1601, 409, 411, 160]),
              ("numbers", 2, [570]), ("goforward", 2, [410, 160, 160, 250]), ("numbers", 2, [16000])]


def livefeat_only():
    """What a LIVE decoder's acoustic front half hands its searches (ref_dump livefeat: chunked acmod_process_raw -- overflow samples,
    cepstrum buffer pieces, cmn_live, the feature window -- over successive utterances of one decoder), for lists of chunk sizes,
    with the growing feature buffer of -fwdflat yes and without: the hash of every feature frame, the frame counts per
    utterance, the running mean after each, and the first case's frames in full."""
    out = {}
    for gi, grow in enumerate(("yes", "no")):
        for ci, (name, nutt, chunks) in enumerate(LIVE_CASES):
            d = ref_dump("livefeat", os.path.join(REF, "data", name + ".raw"), nutt, ",".join(str(c) for c in chunks), extra=("fwdflat", grow))
            k = "g%d_c%d_" % (gi, ci)
            out[k + "hash"] = row_hash(d["feat"]); out[k + "utt_frames"] = d["utt_frames"]; out[k + "mean"] = d["cmn_mean_after"]
            out[k + "chunks"] = np.array(chunks, np.int32); out[k + "nutt"] = np.array([nutt], np.int32)
            out[k + "clip"] = np.frombuffer(name.encode(), np.uint8)
            if ci == 0:
                out[k + "feat"] = d["feat"]
            print("livefeat", grow, name, chunks, list(d["utt_frames"]))
    np.savez_compressed(os.path.join(GOLD, "livefeat_en_us.npz"), **out)
    print(os.path.getsize(os.path.join(GOLD, "livefeat_en_us.npz")))


def fwdtree_session():
    """the SECOND utterance of a session (REFDUMP_WARMUP: the decoder decodes another utterance first): the permanent
    multiplexed channels start with the per-state ssids the first one left (hmm_clear keeps them, hmm.c:181-196) --
    `mpx_init` -- which changes the senones the search lists and so every score's normaliser"""
    base = ("fwdflat", "no", "bestpath", "no")
    for name, warm, audio in (("goforward_after_numbers", "numbers.raw", "goforward.raw"),
                              ("numbers_after_something", "something.raw", "numbers.raw")):
        os.environ["REFDUMP_WARMUP"] = os.path.join(REF, "data", warm)
        try:
            d = ref_dump("fwdtree", os.path.join(REF, "data", audio), extra=base)
        finally:
            del os.environ["REFDUMP_WARMUP"]
        tr = {k: v for k, v in d.items() if k not in FT_STATIC}
        tr["static"] = np.frombuffer(b"en_us_turtle", np.uint8)
        np.savez_compressed(os.path.join(GOLD, "fwdtree_trace_%s.npz" % name), **tr)
        fresh = np.load(os.path.join(GOLD, "fwdtree_trace_%s.npz" % audio.split(".")[0]))
        print("fwdtree session", name, "bp", d["bp"].shape[0], "mpx_init non-BAD beyond state 0:",
              int((d["mpx_init"][:, 1:] != 0xffff).sum()), "| same table as the fresh decode:", np.array_equal(fresh["bp"], d["bp"]),
              "same scores handed:", np.array_equal(fresh["step_scr"], d["step_scr"]) if fresh["step_scr"].shape == d["step_scr"].shape else False)


FF_STATIC = ["pron_off", "pron_ci", "pron_ssid", "ci_ssid", "lm_known"]
FF_DROP = ["step_frame", "step_best", "step_lpbest", "step_bpidx", "step_pen", "step_act_off", "step_act", "step_scr", "step_rest",
           "n_steps"]


def fwdflat_only():
    """Second-pass goldens (ref_dump fwdflat): what the flat-lexicon search adds to a fixture's static tables
    (pronunciations as word-internal ssids, CI ssids, LM membership), and per two-pass decode what pass 1 handed
    over (its back-pointer table, the single-phone channels' ssids), the senone scores every pass-2 frame was
    handed, and the back-pointer table pass 2 produced."""
    tdm = os.path.join(REF, "model", "tidigits")
    tdl = os.path.join(REF, "data", "tidigits")
    base = ("fwdflat", "yes", "bestpath", "no")
    med = dict(lm=os.path.join(REF, "data", "medium.arpa"), dic=os.path.join(REF, "data", "medium.dic"))
    cases = [
        ("en_us_turtle", "goforward", dict(), "goforward.raw", base),
        ("en_us_turtle", "numbers", dict(), "numbers.raw", base),
        ("en_us_turtle", "something_efwid2_sfwin8", dict(), "something.raw",
         base + ("fwdflatefwid", "2", "fwdflatsfwin", "8", "fwdflatbeam", "1e-40", "fwdflatwbeam", "1e-5", "fwdflatlw", "6.0")),
        ("tidigits", "man_ah_2934za", dict(model=tdm, lm=os.path.join(tdl, "tidigits.lm.bin"), dic=os.path.join(tdl, "tidigits.dic")),
         os.path.join("tidigits", "man.ah.2934za.mfc"), base),
        ("en_us_medium", "medium_numbers", med, "numbers.raw", base),
    ]
    done = set()
    for static, name, kw, audio, extra in cases:
        d = ref_dump("fwdflat", os.path.join(REF, "data", audio), extra=extra, **kw)
        if static not in done:
            # a fixture without a dense LM table carries the trie tables of the model THIS decode used (the staged
            # medium.arpa need not be the one fwdtree_static_en_us_medium.npz was made from)
            keys = FF_STATIC + ([k for k in LM_KEYS if k in d] if "lm" not in d else [])
            np.savez_compressed(os.path.join(GOLD, "fwdflat_static_%s.npz" % static), **{k: d[k] for k in keys})
            done.add(static)
        drop = set(FT_STATIC) | set(LM_KEYS) | set(FF_STATIC) | set(FF_DROP)
        tr = {k: v for k, v in d.items() if k not in drop}
        tr["static"] = np.frombuffer(static.encode(), np.uint8)
        path = os.path.join(GOLD, "fwdflat_trace_%s.npz" % name)
        np.savez_compressed(path, **tr)
        print("fwdflat", name, "frames", int(d["flat_n_steps"][0]), "bp1", d["bp1"].shape[0], "bp", d["bp"].shape[0],
              "hyp", bytes(d["hyp"]).decode(), os.path.getsize(path))


def read_arpa(path):
    """n-grams of an ARPA file as lists of word tuples per order"""
    import bz2
    op = bz2.open if path.endswith(".bz2") else open
    grams, order = {}, 0
    with op(path, "rt") as fh:
        for line in fh:
            line = line.strip()
            if line.startswith("\\") and line.endswith("-grams:"):
                order = int(line[1:line.index("-")])
                grams[order] = []
            elif line.startswith("\\end\\"):
                break
            elif order and line:
                f = line.split()
                grams[order].append(tuple(f[1:1 + order]))
    return grams


def write_synthetic_arpa(path, n_words, n_bg, n_tg, seed):
    """a consistent random trigram model (every trigram's two bigrams and every bigram's unigrams exist):
    large enough that the interpolation search of the trie takes several steps"""
    rng = np.random.default_rng(seed)
    words = ["<s>", "</s>"] + ["w%05d" % i for i in range(n_words - 2)]
    bg = set()
    while len(bg) < n_bg:
        a, b = (int(x) for x in rng.integers(0, n_words, 2))
        if a != 1 and b != 0:
            bg.add((a, b))
    succ = {}
    for a, b in bg:
        succ.setdefault(a, []).append(b)
    bgl = sorted(bg)
    tg = set()
    while len(tg) < n_tg:
        a, b = bgl[int(rng.integers(0, len(bgl)))]
        if b in succ:
            tg.add((a, b, succ[b][int(rng.integers(0, len(succ[b])))]))
    with open(path, "w") as fh:
        fh.write("\\data\\\nngram 1=%d\nngram 2=%d\nngram 3=%d\n" % (n_words, len(bg), len(tg)))
        fh.write("\n\\1-grams:\n")
        for w in words:
            fh.write("%.4f %s %.4f\n" % (-99.0 if w == "<s>" else -rng.uniform(1.0, 5.0), w, -rng.uniform(0.0, 1.5)))
        fh.write("\n\\2-grams:\n")
        for a, b in bgl:
            fh.write("%.4f %s %s %.4f\n" % (-rng.uniform(0.05, 4.0), words[a], words[b], -rng.uniform(0.0, 1.0)))
        fh.write("\n\\3-grams:\n")
        for a, b, c in sorted(tg):
            fh.write("%.4f %s %s %s\n" % (-rng.uniform(0.05, 3.0), words[a], words[b], words[c]))
        fh.write("\n\\end\\\n")
    return [tuple(words[i] for i in g) for g in bgl], [tuple(words[i] for i in g) for g in sorted(tg)]


def lm_case(name, cmd, args, grams, seed, n_random=4000, n_scan=24, max_listed=100000, **kw):
    """two runs of `ref_dump lm`: the first for the word list, the second with the queries:
    every listed n-gram (in look-up order: ARPA "a b c" = ngram_tg_score(c, b, a)), shortened and
    perturbed variants, random triples, -1 histories, and full successor scans of a few histories"""
    d = ref_dump(cmd, "-", *args, **kw)
    words = bytes(d["words"]).decode().split("\n")[:-1]
    n = len(words)
    wid = {}
    for i, w in enumerate(words):
        wid.setdefault(w, i)
        wid.setdefault(w.upper(), i)       # (test_ngram/turtle.lm is upper case, the dictionary is not)
    rng = np.random.default_rng(seed)
    q = []
    grams = {o: ([g[int(i)] for i in rng.choice(len(g), max_listed, replace=False)] if len(g) > max_listed else g)
             for o, g in grams.items()}
    for g in grams.get(3, []):
        if all(w in wid for w in g):
            a, b, c = (wid[w] for w in g)
            q += [(c, b, a), (c, b, int(rng.integers(0, n))), (c, b, -1), (int(rng.integers(0, n)), b, a)]
    for g in grams.get(2, []):
        if all(w in wid for w in g):
            a, b = (wid[w] for w in g)
            q += [(b, a, -1), (b, a, int(rng.integers(0, n))), (b, -1, a)]
    for i in range(n):
        q.append((i, -1, -1))
    r = rng.integers(-1, n, (n_random, 3)); r[:, 0] = np.abs(r[:, 0])
    q += [tuple(int(x) for x in t) for t in r]
    tgs = [g for g in grams.get(3, []) if all(w in wid for w in g)]
    for k in range(n_scan):
        if tgs and k % 2 == 0:
            g = tgs[int(rng.integers(0, len(tgs)))]; h = (wid[g[1]], wid[g[0]])
        else:
            h = (int(rng.integers(0, n)), int(rng.integers(-1, n)))
        w3 = np.arange(n) if n <= 512 else rng.integers(0, n, 512)
        q += [(int(w), h[0], h[1]) for w in w3]
    qa = np.asarray(q, np.int32)
    with tempfile.NamedTemporaryFile(suffix=".q", delete=False) as fh:
        fh.write(qa.tobytes()); qf = fh.name
    d = ref_dump(cmd, qf, *args, **kw)
    os.unlink(qf)
    assert np.array_equal(d["queries"], qa)
    np.savez_compressed(os.path.join(GOLD, "lm_%s.npz" % name), **d)
    print("lm", name, "order", int(d["order"][0]), "words", n, "queries", len(qa),
          "n_used histogram", np.bincount(d["n_used"], minlength=4).tolist(),
          "bytes", os.path.getsize(os.path.join(GOLD, "lm_%s.npz" % name)))


def lm_only():
    """Language-model goldens: the trie's tables and the reference's ngram_tg_score answers."""
    tn = "/root/reference/test/unit/test_ngram"
    # the model of test/unit/test_ngram/test_lm_score.c, with the weights that test applies (7.5, 0.5)
    lm_case("100", "lm_file", (7.5, 0.5), read_arpa(os.path.join(tn, "100.lm.bz2")), 11,
            lm=os.path.join(tn, "100.lm.bin"), model="-", dic="-")
    # the decoder's own model set: dictionary word ids, -lw / -wip of the default configuration
    lm_case("turtle_decoder", "lm", (), read_arpa(os.path.join(tn, "turtle.lm")), 12)
    # the second search fixture's model (tests cross-check the whole dense table of fwdtree_static_tidigits)
    tdl = os.path.join(REF, "data", "tidigits")
    lm_case("tidigits_decoder", "lm", (), {}, 15, model=os.path.join(REF, "model", "tidigits"),
            lm=os.path.join(tdl, "tidigits.lm.bin"), dic=os.path.join(tdl, "tidigits.dic"))
    # a random consistent trigram model, read from ARPA text: deeper interpolation searches
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "synth.arpa")
        bg, tg = write_synthetic_arpa(path, 3000, 30000, 40000, 13)
        lm_case("synthetic", "lm_file", (6.5, 0.65), {2: bg, 3: tg}, 14, n_scan=8, max_listed=4000, lm=path, model="-", dic="-")


LM_KEYS = ["order", "n_unigrams", "n_words", "unigrams", "ngram_mem", "levels", "quant", "lw", "log_wip", "log_zero",
           "widmap", "words"]


def fwdtree_medium_only():
    """The same goldens for a 715-word task (oracle/make_medium_task.py, staged by `make -C oracle`): too many
    words for the dense LM table, so the static file carries the model's trie tables (LM_KEYS) instead."""
    base = ("fwdflat", "no", "bestpath", "no")
    kw = dict(lm=os.path.join(REF, "data", "medium.arpa"), dic=os.path.join(REF, "data", "medium.dic"))
    static = "en_us_medium"
    for i, (name, audio, extra) in enumerate([("medium_goforward", "goforward.raw", base),
                                              ("medium_numbers_maxwpf8", "numbers.raw", base + ("maxwpf", "8", "maxhmmpf", "1500"))]):
        d = ref_dump("fwdtree", os.path.join(REF, "data", audio), extra=extra, **kw)
        keys = [k for k in FT_STATIC if k != "lm"] + LM_KEYS
        if i == 0:
            np.savez_compressed(os.path.join(GOLD, "fwdtree_static_%s.npz" % static), **{k: d[k] for k in keys})
        tr = {k: v for k, v in d.items() if k not in keys}
        tr["static"] = np.frombuffer(static.encode(), np.uint8)
        np.savez_compressed(os.path.join(GOLD, "fwdtree_trace_%s.npz" % name), **tr)
        print("fwdtree", name, "steps", int(d["n_steps"][0]), "bp", d["bp"].shape[0], "hyp", bytes(d["hyp"]).decode(),
              os.path.getsize(os.path.join(GOLD, "fwdtree_trace_%s.npz" % name)))
    print("static", os.path.getsize(os.path.join(GOLD, "fwdtree_static_%s.npz" % static)))


def hmm_syn_case(n_emit, n_hmm, n_steps, seed):
    """a synthetic context with n_emit emitting states (ref_dump hmmsyn): 1, 2 and 4 reach hmm_vit_eval_anytopo"""
    with tempfile.NamedTemporaryFile(suffix=".psgb", delete=False) as fh:
        out = fh.name
    subprocess.check_call([os.path.join(REF, "ref_dump"), "hmmsyn", out, str(n_emit), str(n_hmm), str(n_steps), str(seed)])
    d = read_psgb(out)
    os.unlink(out)
    np.savez_compressed(os.path.join(GOLD, "hmm_syn_%dst.npz" % n_emit), **d)
    print("hmm_syn_%dst: n_hmm=%d steps=%d" % (n_emit, n_hmm, n_steps))


def hmm_only():
    for ne, seed in ((4, 41), (2, 42), (1, 43)):
        hmm_syn_case(ne, 768, 10, seed)
    # 3-state (en-us) and 5-state (tidigits) topologies, mpx and non-mpx
    hmm_case("en_us_3st", MODEL, LM, DIC, 1536, 12, 20260922)
    tdm = os.path.join(REF, "model", "tidigits")
    tdl = os.path.join(REF, "data", "tidigits")
    hmm_case("tidigits_5st", tdm, os.path.join(tdl, "tidigits.lm.bin"), os.path.join(tdl, "tidigits.dic"),
             1536, 12, 20260923)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "hmm":
        os.makedirs(GOLD, exist_ok=True)
        hmm_only()
    elif len(sys.argv) > 1 and sys.argv[1] == "session":
        fwdtree_session()
    elif len(sys.argv) > 1 and sys.argv[1] == "ptm4":
        ptm_4bit()
    elif len(sys.argv) > 1 and sys.argv[1] == "semi":
        semi_only()
    elif len(sys.argv) > 1 and sys.argv[1] == "ms":
        ms_only()
    elif len(sys.argv) > 1 and sys.argv[1] == "fwdtree":
        fwdtree_only()
    elif len(sys.argv) > 1 and sys.argv[1] == "fwdtree_medium":
        fwdtree_medium_only()
    elif len(sys.argv) > 1 and sys.argv[1] == "fwdflat":
        fwdflat_only()
    elif len(sys.argv) > 1 and sys.argv[1] == "lm":
        lm_only()
    elif len(sys.argv) > 1 and sys.argv[1] == "mfcc":
        mfcc_only()
    elif len(sys.argv) > 1 and sys.argv[1] == "dynfeat":
        dynfeat_only()
    elif len(sys.argv) > 1 and sys.argv[1] == "lm_set":
        lm_set_only()
    elif len(sys.argv) > 1 and sys.argv[1] == "feat_types":
        feat_types_only()
    elif len(sys.argv) > 1 and sys.argv[1] == "livefeat":
        livefeat_only()
    elif len(sys.argv) > 1 and sys.argv[1] == "fwdtree_topn":
        fwdtree_topn_only()
    elif len(sys.argv) > 1 and sys.argv[1] == "ptm_topn":
        ptm_topn_only()
    else:
        main()

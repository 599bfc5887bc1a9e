"""The large-vocabulary decode task (en-us PTM + the synthetic 126k-unigram LM over every entry of
cmudict-en-us.dict, 134,865 dictionary words, 248 k lexicon-tree channels) as a device pipeline.

The search's static tables -- lexicon tree, dictionary columns, dict2pid maps, beams, the LM's trie -- are the reference's own:
read out of a decoder the compiled reference initialised (oracle/_ref/ref_dump fwdtree, the same dump the search goldens
are made from; ~20 s).  That is model loading, exactly what integration/psgpu_device_decode.c does through pointers inside a
live process; nothing of the reference runs in the decode itself.  (reference: ngram_fwdtree_init,
src/ngram_search_fwdtree.c:66-336; dict2pid_build, src/dict2pid.c; lm_trie, src/lm/lm_trie.c.)"""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref")
LM = os.path.join(REF, "data", "big.arpa")
DICT = os.path.join(REF, "data", "cmudict-en-us.dict")

# back-pointers / score-stack entries per frame the benchmark's synthetic utterances write at this vocabulary (measured with
# the reference: 25-30 and 650-800); the pipeline doubles its tables when an utterance needs more (psgpu_decode_table_capacity)
BP_PER_FRAME, BSS_PER_FRAME = 48, 1200


def available():
    return all(os.path.exists(p) for p in (os.path.join(REF, "ref_dump"), LM, DICT, os.path.join(REF, "model", "en-us", "mdef")))


def tables(out_dir=None, two_pass=False):
    """dict of the task's tables (numpy arrays): the `ref_dump fwdtree` record of a decoder initialised with big.arpa +
    cmudict-en-us.dict, -fwdflat no -bestpath no (it also holds that decoder's trace of goforward.raw: a golden).
    two_pass: the `ref_dump fwdflat` record of a -fwdflat yes decoder instead -- the same tables plus what the flat-lexicon
    second pass adds (pronunciations, its beams, the language-weight ratio) and the two-pass trace of goforward.raw."""
    if not available():
        raise RuntimeError("the large-vocabulary task needs oracle/_ref (ref_dump + staged model, big.arpa, cmudict): make -C oracle ref")
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from psgb import read_psgb
    tmp = out_dir or tempfile.mkdtemp(prefix="psgpu_big_")
    out = os.path.join(tmp, "big.psgb")
    subprocess.check_call([os.path.join(REF, "ref_dump"), "fwdflat" if two_pass else "fwdtree", out, os.path.join(REF, "model", "en-us"), LM, DICT,
                           os.path.join(REF, "data", "goforward.raw"), "--", "fwdflat", "yes" if two_pass else "no", "bestpath", "no"],
                          stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=900)
    g = read_psgb(out)
    if out_dir is None:
        os.unlink(out); os.rmdir(tmp)
    return g


def words_of(g):
    """the dictionary's word strings by word id"""
    return bytes(g["words"]).decode().split("\n")


def pipeline(g, fe_tables, ptm_tables, max_words=512):
    """DecodePipeline on the task's tables with the LM's trie on the device"""
    from . import DecodePipeline, NGramTrieLM
    lm = NGramTrieLM(g)
    p = DecodePipeline(fe_tables, ptm_tables, g, g["par"], g, lm=lm, max_words=max_words)
    p._lm = lm
    p.table_capacity(BP_PER_FRAME, BSS_PER_FRAME, True)
    return p

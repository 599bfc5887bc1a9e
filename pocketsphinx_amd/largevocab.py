"""The large-vocabulary decode task (en-us PTM + the synthetic 126k-unigram LM over every entry of
cmudict-en-us.dict, 134,865 dictionary words, 248 k lexicon-tree channels) as a device pipeline.

The search's static tables -- lexicon tree, dictionary columns, dict2pid maps, beams, the LM's trie, and what the flat-lexicon
second pass adds -- come from a TABLE FILE written by the reference-side tool integration/psgpu_export_tables (a decoder
initialised as any application does, flattened by the binding's own code integration/psgpu_search_tables.c, the same that
psgpu_device_search_attach runs on a live decoder).  `make -C integration tables` (part of __graft_entry__.build()) writes
the files this module looks for by default.  That is model loading; nothing of the reference runs in the decode.
(reference: ngram_fwdtree_init, src/ngram_search_fwdtree.c:66-336; dict2pid_build, src/dict2pid.c; lm_trie, src/lm/lm_trie.c.)"""
import os

from .tablefile import read_psgb

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TASK = "en_us_cmudict_big"

# back-pointers / score-stack entries per frame the benchmark's synthetic utterances write at this vocabulary (measured with
# the reference: 25-30 and 650-800); the pipeline doubles its tables when an utterance needs more (psgpu_decode_table_capacity)
BP_PER_FRAME, BSS_PER_FRAME = 48, 1200


def table_dir():
    """PSGPU_TABLE_DIR, or where `make -C integration tables` writes"""
    return os.environ.get("PSGPU_TABLE_DIR") or os.path.join(ROOT, "integration", "_tables")


def table_path(task=TASK, directory=None):
    return os.path.join(directory or table_dir(), task + ".psgb")


def available(path=None):
    return os.path.exists(path or table_path())


def tables(path=None):
    """dict of the task's tables (numpy arrays) from a table file (default: table_path()).  The file of a decoder configured
    with -fwdflat yes also holds what the second pass needs (pronunciations, its beams, the language-weight ratio)."""
    path = path or table_path()
    if not os.path.exists(path):
        raise RuntimeError("table file %s not found: write it with integration/psgpu_export_tables (`make -C integration tables`) "
                           "or set PSGPU_TABLE_DIR" % path)
    return read_psgb(path)


def words_of(g):
    """the dictionary's word strings by word id"""
    return bytes(g["dict_words"] if "dict_words" in g else g["words"]).decode().split("\n")


def pipeline(g, fe_tables, ptm_tables, max_words=512):
    """DecodePipeline on the task's tables with the LM's trie on the device"""
    from . import DecodePipeline, NGramTrieLM
    lm = NGramTrieLM(g)
    p = DecodePipeline(fe_tables, ptm_tables, g, g["par"], g, lm=lm, max_words=max_words)
    p._lm = lm
    p.table_capacity(BP_PER_FRAME, BSS_PER_FRAME, True)
    return p

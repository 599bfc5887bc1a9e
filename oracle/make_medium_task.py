#!/usr/bin/env python3
"""TEST INFRASTRUCTURE.  A medium-vocabulary task for the device first pass: the
turtle dictionary plus several hundred cmudict words, with a random but consistent trigram model
over them (ARPA text, which the reference reads directly) -- a vocabulary the dense LM table of the
first search version could not hold, a lexicon tree of a few thousand nodes, and a trie deep enough
that its interpolation search takes several steps.

usage: make_medium_task.py CMUDICT TURTLE.dic OUT.dic OUT.arpa [N_EXTRA_WORDS]
"""
import random
import sys


def read_dict(path):
    out = []
    for line in open(path, encoding="utf-8", errors="replace"):
        f = line.split()
        if len(f) >= 2:
            out.append((f[0], f[1:]))
    return out


def main():
    cmu, turtle, out_dic, out_lm = sys.argv[1:5]
    n_extra = int(sys.argv[5]) if len(sys.argv) > 5 else 600
    rnd = random.Random(20260923)
    entries = read_dict(turtle)
    have = {w for w, _ in entries}
    cand = [(w, p) for w, p in read_dict(cmu) if "(" not in w and w.isalpha() and w not in have and 2 <= len(p) <= 9]
    step = max(1, len(cand) // n_extra)
    entries += cand[::step][:n_extra]
    with open(out_dic, "w") as fh:
        for w, p in entries:
            fh.write("%s %s\n" % (w, " ".join(p)))
    words = sorted({w.split("(")[0] for w, _ in entries})
    vocab = ["<s>", "</s>"] + words
    n = len(vocab)
    bg = {("<s>", "go"), ("go", "forward"), ("forward", "ten"), ("ten", "meters"), ("meters", "</s>")}
    while len(bg) < 6 * n:
        a, b = vocab[rnd.randrange(n)], vocab[rnd.randrange(n)]
        if a != "</s>" and b != "<s>":
            bg.add((a, b))
    bgl = sorted(bg)
    succ = {}
    for a, b in bgl:                   # (sorted: iterating the set itself made the model depend on PYTHONHASHSEED)
        succ.setdefault(a, []).append(b)
    tg = {("<s>", "go", "forward"), ("go", "forward", "ten"), ("forward", "ten", "meters"), ("ten", "meters", "</s>")}
    while len(tg) < 5 * n:
        a, b = bgl[rnd.randrange(len(bgl))]
        if b in succ:
            tg.add((a, b, succ[b][rnd.randrange(len(succ[b]))]))
    with open(out_lm, "w") as fh:
        fh.write("\\data\\\nngram 1=%d\nngram 2=%d\nngram 3=%d\n\n\\1-grams:\n" % (n, len(bgl), len(tg)))
        for w in vocab:
            lp = -99.0 if w == "<s>" else (-1.5 if w in ("go", "forward", "ten", "meters", "</s>") else -rnd.uniform(2.5, 4.5))
            fh.write("%.4f %s %.4f\n" % (lp, w, -rnd.uniform(0.1, 0.9)))
        fh.write("\n\\2-grams:\n")
        for a, b in bgl:
            fh.write("%.4f %s %s %.4f\n" % (-rnd.uniform(0.2, 2.5), a, b, -rnd.uniform(0.0, 0.7)))
        fh.write("\n\\3-grams:\n")
        for a, b, c in sorted(tg):
            fh.write("%.4f %s %s %s\n" % (-rnd.uniform(0.1, 2.0), a, b, c))
        fh.write("\n\\end\\\n")
    print("medium task: %d dictionary entries, %d LM words, %d bigrams, %d trigrams" % (len(entries), n, len(bgl), len(tg)))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""TEST INFRASTRUCTURE.  Synthetic large-vocabulary LM for the drop-in tests
(SURVEY F9b / 8d config 3): the reference ships no large LM (en-us.lm.bin is
absent), so the search is exercised at realistic scale -- ~250k lextree
channels, ~9,000 active HMMs per frame -- with unigrams over every base word of
cmudict-en-us.dict (Zipf-like weights, a few boosted words) plus a handful of
bigrams, written as an ARPA file the reference reads directly.

usage: make_biglm.py CMUDICT OUT.arpa
"""
import math
import sys


def main():
    dic, out = sys.argv[1], sys.argv[2]
    words = []
    seen = set()
    for line in open(dic, encoding="utf-8", errors="replace"):
        w = line.split(None, 1)[0] if line.strip() else ""
        if not w or "(" in w or w in seen or w in ("<s>", "</s>", "<sil>"):
            continue
        seen.add(w)
        words.append(w)
    boost = {"go": 0.02, "forward": 0.02, "ten": 0.02, "meters": 0.02, "one": 0.01, "two": 0.01, "the": 0.03}
    z = [1.0 / (i + 10.0) for i in range(len(words))]
    zs = sum(z)
    rest = 1.0 - sum(boost.values()) - 0.05          # 0.05 for </s>
    p = {w: rest * zi / zs for w, zi in zip(words, z)}
    for w, b in boost.items():
        if w in p:
            p[w] += b
    bigrams = [("<s>", "go", -0.3), ("go", "forward", -0.2), ("forward", "ten", -0.3),
               ("ten", "meters", -0.2), ("meters", "</s>", -0.2)]
    with open(out, "w") as fh:
        fh.write("\\data\\\nngram 1=%d\nngram 2=%d\n\n\\1-grams:\n" % (len(words) + 2, len(bigrams)))
        fh.write("-99.0000 <s> -0.3010\n")
        fh.write("%.4f </s>\n" % math.log10(0.05))
        for w in words:
            fh.write("%.4f %s -0.3010\n" % (math.log10(p[w]), w))
        fh.write("\n\\2-grams:\n")
        for a, b, lp in bigrams:
            fh.write("%.4f %s %s\n" % (lp, a, b))
        fh.write("\n\\end\\\n")
    print("big LM: %d words -> %s" % (len(words), out))


if __name__ == "__main__":
    main()

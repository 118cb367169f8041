"""BLEU-1..4 and ROUGE-L corpus scorers for the dense-captioning evaluation (reference:
lib/capeval/bleu/bleu_scorer.py:23-264 with the "closest" reference length that lib/capeval/bleu/bleu.py:39 selects,
lib/capeval/rouge/rouge.py:13-100; called from lib/captioning/eval_helper.py:289-291).  Host-side string work as in the
reference; the n-gram counter is shared with d3net_amd.cider.  Same accumulation order, so the float64 results equal the
reference's.  METEOR (a Java subprocess in the reference) is not provided.
"""
import math

import numpy as np

from .cider import ngram_counts

_SMALL, _TINY = 1e-9, 1e-15


def bleu_scores(references, candidates, n=4):
    """references: list of lists of sentences, candidates: list of sentences ->
    ([corpus BLEU-1..n], [per-entry BLEU-k lists]) == Bleu(n).compute_score(gts, res)"""
    assert len(references) == len(candidates)
    per = [[] for _ in range(n)]
    tot_guess, tot_correct, tot_test, tot_ref = [0] * n, [0] * n, 0, 0
    for refs, cand in zip(references, candidates):
        maxc, reflens = {}, []
        for r in refs:
            reflens.append(len(r.split()))
            for g, c in ngram_counts(r, n).items():
                if c > maxc.get(g, 0):
                    maxc[g] = c
        testlen = len(cand.split())
        reflen = min((abs(l - testlen), l) for l in reflens)[1]            # "closest"
        guess = [max(0, testlen - k + 1) for k in range(1, n + 1)]
        correct = [0] * n
        for g, c in ngram_counts(cand, n).items():
            correct[len(g) - 1] += min(maxc.get(g, 0), c)
        tot_test += testlen; tot_ref += reflen
        b = 1.0
        for k in range(n):
            tot_guess[k] += guess[k]; tot_correct[k] += correct[k]
            b *= (float(correct[k]) + _TINY) / (float(guess[k]) + _SMALL)
            per[k].append(b ** (1.0 / (k + 1)))
        ratio = (testlen + _TINY) / (reflen + _SMALL)
        if ratio < 1:
            for k in range(n):
                per[k][-1] *= math.exp(1 - 1 / ratio)
    bleus, b = [], 1.0
    for k in range(n):
        b *= float(tot_correct[k] + _TINY) / (tot_guess[k] + _SMALL)
        bleus.append(b ** (1.0 / (k + 1)))
    ratio = (tot_test + _TINY) / (tot_ref + _SMALL)
    if ratio < 1:
        bleus = [x * math.exp(1 - 1 / ratio) for x in bleus]
    return bleus, per


def _lcs(a, b):
    """length of the longest common subsequence of two token lists (two-row dynamic programme)"""
    if len(a) < len(b):
        a, b = b, a
    prev = [0] * (len(b) + 1)
    for x in a:
        cur = [0]
        for j, y in enumerate(b, 1):
            cur.append(prev[j - 1] + 1 if x == y else max(prev[j], cur[j - 1]))
        prev = cur
    return prev[len(b)]


def rouge_l_scores(references, candidates, beta=1.2):
    """-> (mean ROUGE-L, per-entry float64 array) == Rouge().compute_score(gts, res); tokens split on single spaces, as
    the reference does (rouge.py:58,62)"""
    scores = []
    for refs, cand in zip(references, candidates):
        tc = cand.split(" ")
        prec, rec = [], []
        for r in refs:
            tr = r.split(" ")
            l = _lcs(tr, tc)
            prec.append(l / float(len(tc))); rec.append(l / float(len(tr)))
        p, r = max(prec), max(rec)
        scores.append(((1 + beta ** 2) * p * r) / float(r + beta ** 2 * p) if p != 0 and r != 0 else 0.0)
    scores = np.array(scores)
    return float(np.mean(scores)), scores

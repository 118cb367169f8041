"""CIDEr reward scorer for the self-critical speaker update (reference: lib/capeval/cider/cider.py:13-55 and
cider_scorer.py:11-193, used by lib/captioning/loss_helper.py:15-96).

Host-side string work, as in the reference -- but organised for the way the reward is called: the same reference
sentence sets recur once per sampled beam and per chunk entry, so every distinct sentence is n-gram counted and tf-idf
weighted ONCE per call (interned by string), and a (candidate, reference-set) pair that recurs is scored once.
Accumulation orders follow the reference so the float64 scores are bit-identical to its own:
  * n-gram insertion order = order 1..4, then position (precook, cider_scorer.py:11-27);
  * `length` counts the BIGRAMS of a sentence (the `n == 1` test at cider_scorer.py:127 is on the 0-based order);
  * the document frequency of an n-gram = number of (key) entries whose reference SET contains it, duplicates of the
    same set counted each time (compute_doc_freq, :92-103); log reference length = log(#entries) (:160).
"""
import math

import numpy as np

N_ORDERS = 4
SIGMA = 6.0


def ngram_counts(sentence, n=N_ORDERS):
    """sentence string -> {ngram tuple: count}, keys in the reference's insertion order"""
    words = sentence.split()
    counts = {}
    for k in range(1, n + 1):
        for i in range(len(words) - k + 1):
            g = tuple(words[i:i + k])
            counts[g] = counts.get(g, 0) + 1
    return counts


class _Vec:
    """tf-idf vector of one sentence: per order a dict ngram -> weight, its L2 norm, and the bigram count"""
    __slots__ = ("w", "norm", "length")

    def __init__(self, counts, doc_freq, log_ref_len, n):
        self.w = [dict() for _ in range(n)]
        sq = [0.0] * n
        self.length = 0
        for g, tf in counts.items():
            df = np.log(max(1.0, doc_freq.get(g, 0.0)))
            o = len(g) - 1
            v = float(tf) * (log_ref_len - df)
            self.w[o][g] = v
            sq[o] += pow(v, 2)
            if o == 1:
                self.length += tf
        self.norm = [np.sqrt(x) for x in sq]


def _similarity(hyp, ref, n, sigma):
    """clipped cosine similarity per n-gram order with the gaussian length penalty (cider_scorer.py:135-157)"""
    delta = float(hyp.length - ref.length)
    penalty = np.e ** (-(delta ** 2) / (2 * sigma ** 2))
    val = np.zeros(n)
    for o in range(n):
        rw = ref.w[o]
        acc = 0.0
        for g, hv in hyp.w[o].items():
            rv = rw.get(g, 0.0)
            acc += min(hv, rv) * rv
        if hyp.norm[o] != 0 and ref.norm[o] != 0:
            acc /= (hyp.norm[o] * ref.norm[o])
        assert not math.isnan(acc)
        val[o] = acc * penalty
    return val


def cider_scores(references, candidates, n=N_ORDERS, sigma=SIGMA):
    """references: list (one per entry) of lists of reference sentences; candidates: list of candidate sentences.
    Returns (mean score, float64 array of per-entry scores) == Cider().compute_score(gts, res) with
    gts[str(i)] = references[i], res[str(i)] = [candidates[i]]."""
    assert len(references) == len(candidates) and len(candidates) > 0
    counted = {}

    def counts_of(s):
        c = counted.get(s)
        if c is None:
            c = counted[s] = ngram_counts(s, n)
        return c

    # document frequency over entries (an entry = one reference set)
    doc_freq = {}
    set_cache = {}
    for refs in references:
        assert len(refs) > 0
        key = tuple(refs)
        grams = set_cache.get(key)
        if grams is None:
            grams = set()
            for r in refs:
                grams.update(counts_of(r))
            set_cache[key] = grams
        for g in grams:
            doc_freq[g] = doc_freq.get(g, 0.0) + 1
    log_ref_len = np.log(float(len(references)))

    vecs = {}

    def vec_of(s):
        v = vecs.get(s)
        if v is None:
            v = vecs[s] = _Vec(counts_of(s), doc_freq, log_ref_len, n)
        return v

    pair_cache = {}
    scores = np.zeros(len(candidates))
    for i, (cand, refs) in enumerate(zip(candidates, references)):
        key = (cand, tuple(refs))
        s = pair_cache.get(key)
        if s is None:
            hv = vec_of(cand)
            acc = np.zeros(n)
            for r in refs:
                acc += _similarity(hv, vec_of(r), n, sigma)
            s = np.mean(acc)
            s /= len(refs)
            s *= 10.0
            pair_cache[key] = s
        scores[i] = s
    return float(np.mean(scores)), scores

"""Synthetic corpora of the BASELINE.json configurations.

There is no dataset to download (the reference's real inputs are private NAPKON files), so
every benchmark and parity test runs on these generators.  All are seeded and vectorised.
"""
from __future__ import annotations

from typing import List, Tuple

import numpy as np

STRING_ALPHABET = "abcdefghijklmnopqrstuvwxyz0123456789 "  # fixed point of default_process
SPACE_CODE = len(STRING_ALPHABET) - 1


# ------------------------------------------------------------------ C2 / C4: token-id sets
def token_sets(n: int, seed: int, mean: float = 8.0, width: int = 16, id_range: int = 1 << 17) -> np.ndarray:
    """int32 [n][width], each row a sorted set of clip(Poisson(mean), 1, width) distinct ids
    drawn from U[0, id_range), padded with -1."""
    rng = np.random.default_rng(seed)
    size = np.clip(rng.poisson(mean, n), 1, width).astype(np.int32)
    ids = rng.integers(0, id_range, size=(n, width), dtype=np.int64)
    while True:  # re-draw the (rare) rows that drew one id twice
        srt = np.sort(ids, axis=1)
        bad = (srt[:, 1:] == srt[:, :-1]).any(axis=1)
        if not bad.any():
            break
        ids[bad] = rng.integers(0, id_range, size=(int(bad.sum()), width), dtype=np.int64)
    big = np.iinfo(np.int64).max
    ids[np.arange(width)[None, :] >= size[:, None]] = big
    ids.sort(axis=1)
    ids[ids == big] = -1
    return ids.astype(np.int32)


def plant_near_duplicate_sets(
    left: np.ndarray, right: np.ndarray, seed: int, fraction: float = 0.01, id_range: int = 1 << 17
) -> np.ndarray:
    """Overwrite ``fraction`` of the right rows with copies of random left rows in which at most
    one id was replaced (guarantees above-threshold pairs)."""
    rng = np.random.default_rng(seed)
    right = right.copy()
    n_plant = max(1, int(round(fraction * right.shape[0])))
    targets = rng.choice(right.shape[0], size=n_plant, replace=False)
    sources = rng.integers(0, left.shape[0], size=n_plant)
    for t, s in zip(targets, sources):
        row = left[s].copy()
        k = int((row >= 0).sum())
        if rng.random() < 0.5 and k > 0:
            new = int(rng.integers(0, id_range))
            if new not in row[:k]:
                row[int(rng.integers(0, k))] = new
                vals = np.sort(row[:k])
                row[:k] = vals
        right[t] = row
    return right


def c2_corpus(n: int = 50_000, m: int = 50_000, seed_left: int = 1234, seed_right: int = 5678):
    left = token_sets(n, seed_left)
    right = plant_near_duplicate_sets(left, token_sets(m, seed_right), seed_right + 1)
    return left, right


# ------------------------------------------------------------------ C3: strings
def strings(n: int, seed: int, lo: int = 16, hi: int = 64) -> Tuple[np.ndarray, np.ndarray]:
    """uint8 codes [n][64] over STRING_ALPHABET with length ~ U[lo, hi]; no leading / trailing
    blank (so every string is a fixed point of default_process)."""
    rng = np.random.default_rng(seed)
    length = rng.integers(lo, hi + 1, size=n).astype(np.int32)
    codes = rng.integers(0, len(STRING_ALPHABET), size=(n, 64), dtype=np.int64).astype(np.uint8)
    first = codes[:, 0]
    first[first == SPACE_CODE] = 0
    last = codes[np.arange(n), length - 1]
    last[last == SPACE_CODE] = 1
    codes[np.arange(n), length - 1] = last
    codes[np.arange(64)[None, :] >= length[:, None]] = 0
    return codes, length


def plant_near_duplicate_strings(
    left: Tuple[np.ndarray, np.ndarray], right: Tuple[np.ndarray, np.ndarray], seed: int, fraction: float = 0.01
):
    """Overwrite ``fraction`` of the right strings with copies of random left strings carrying up
    to 10 % random substitutions."""
    rng = np.random.default_rng(seed)
    lc, ll = left
    rc, rl = right[0].copy(), right[1].copy()
    n_plant = max(1, int(round(fraction * rc.shape[0])))
    targets = rng.choice(rc.shape[0], size=n_plant, replace=False)
    sources = rng.integers(0, lc.shape[0], size=n_plant)
    for t, s in zip(targets, sources):
        row, length = lc[s].copy(), int(ll[s])
        for _ in range(int(rng.integers(0, length // 10 + 1))):
            pos = int(rng.integers(1, max(2, length - 1)))
            row[pos] = rng.integers(0, SPACE_CODE)  # never a blank: keeps the fixed point
        rc[t], rl[t] = row, length
    return rc, rl


def c3_corpus(n: int = 200_000, m: int = 200_000, seed_left: int = 1234, seed_right: int = 5678):
    left = strings(n, seed_left)
    right = plant_near_duplicate_strings(left, strings(m, seed_right), seed_right + 1)
    return left, right


def decode_strings(codes: np.ndarray, length: np.ndarray) -> List[str]:
    return ["".join(STRING_ALPHABET[c] for c in row[:k]) for row, k in zip(codes, length)]


def decode_sets(ids: np.ndarray) -> List[List[str]]:
    """Token lists as the reference's plugin would see them (``str`` tokens)."""
    return [[f"t{v}" for v in row if v >= 0] for row in ids]


# ------------------------------------------------------------------ C1 / C5: cohorts
def cohort_records(
    name: str, n: int, seed: int, vocab: int = 500, max_entries: int = 8, tokens_per_entry: int = 1,
    n_categories: int = 8, min_entries: int = 1,
) -> list:
    """hap / pop / suep shaped items: ``Tokens`` is a list of entries, each a blank-joined group
    of ``t<id>`` words, so that gen_comp_value yields one suffix-nested level per entry."""
    rng = np.random.default_rng(seed)
    rows = []
    for k in range(n):
        entries = [
            " ".join(f"t{int(v)}" for v in rng.integers(0, vocab, size=tokens_per_entry))
            for _ in range(int(rng.integers(min_entries, max_entries + 1)))
        ]
        cats = sorted(f"cat{int(c)}" for c in rng.choice(n_categories, size=int(rng.integers(1, 3)), replace=False))
        rows.append(
            {
                "Identifier": f"{name}#sheet{k % 7}#{k}",
                "Variable": f"{name}_var_{k}",
                "Sheet": f"sheet{k % 7}",
                "Category": cats,
                "Term": [f"header{k % 5}", f"question {k}"],
                "Tokens": entries,
                "Parameter": f"param{k}",
            }
        )
    return rows


# ------------------------------------------------------------------ C5: cohorts at scale (vectorised)
def c5_cohort(n: int, seed: int, vocab: int = 20_000, entries: int = 4, tokens_per_entry: int = 2,
              n_categories: int = 32, plant_from: dict = None, plant_fraction: float = 0.01, lex: List[str] = None):
    """hap / pop / suep shaped cohort of BASELINE configs[4]: every item has ``entries`` entries of
    ``tokens_per_entry`` words ``t<id>`` (-> ``entries`` suffix-nested levels, ~8 ids) and 1-2 of
    ``n_categories`` category labels.  Returns a dict with
      tok    int32 [n][entries*tpe]  word ids, entry e at columns [e*tpe, (e+1)*tpe)
      ids    int32 [n][entries*tpe]  the same words in suffix-nested order (last entry first),
                                     de-duplicated, unique ids first, padded with -1
      plen   uint8 [n][entries]      level l = first plen[l] ids
      nlev   int32 [n]
      cat    uint64[n]               category bit mask
    ``lex``: the words behind the ids (``word_vocabulary``; None = ``t<id>``): the "c5w" workload -- configs[4]'s shape on
    word-like text instead of digit strings.
    """
    if lex is not None:
        vocab = len(lex)
    rng = np.random.default_rng(seed)
    width = entries * tokens_per_entry
    tok = rng.integers(0, vocab, size=(n, width), dtype=np.int64).astype(np.int32)
    if plant_from is not None:  # near-duplicate items: copy a source item, re-draw one word
        n_plant = max(1, int(round(plant_fraction * n)))
        targets = rng.choice(n, size=n_plant, replace=False)
        sources = rng.integers(0, plant_from["tok"].shape[0], size=n_plant)
        tok[targets] = plant_from["tok"][sources]
        change = rng.random(n_plant) < 0.5
        tok[targets[change], rng.integers(0, width, size=int(change.sum()))] = rng.integers(
            0, vocab, size=int(change.sum()))
        return _finish_c5(tok, rng, entries, tokens_per_entry, n_categories, vocab, targets, plant_from, sources, lex)
    return _finish_c5(tok, rng, entries, tokens_per_entry, n_categories, vocab, None, None, None, lex)


def _finish_c5(tok, rng, entries, tokens_per_entry, n_categories, vocab, targets, plant_from, sources, lex=None):
    n, width = tok.shape
    cols = np.concatenate([np.arange(e * tokens_per_entry, (e + 1) * tokens_per_entry) for e in range(entries - 1, -1, -1)])
    nested = tok[:, cols]
    first = np.ones((n, width), dtype=bool)
    for p in range(1, width):
        first[:, p] = ~(nested[:, :p] == nested[:, p: p + 1]).any(axis=1)
    seen = np.cumsum(first, axis=1)
    plen = seen[:, tokens_per_entry - 1:: tokens_per_entry].astype(np.uint8)
    order = np.argsort(~first, axis=1, kind="stable")
    ids = np.take_along_axis(np.where(first, nested, -1), order, axis=1).astype(np.int32)
    k = rng.integers(1, 3, size=n)
    c1 = rng.integers(0, n_categories, size=n)
    c2 = rng.integers(0, n_categories, size=n)
    cat = (np.uint64(1) << c1.astype(np.uint64)) | np.where(k == 2, np.uint64(1) << c2.astype(np.uint64), np.uint64(0))
    if targets is not None:  # planted items keep their source's categories (so the pair survives the filter)
        cat[targets] = plant_from["cat"][sources]
    return {"tok": tok, "ids": ids, "plen": plen, "nlev": np.full(n, entries, dtype=np.int32), "cat": cat,
            "entries": entries, "tokens_per_entry": tokens_per_entry, "lex": lex}


def c5_level_token_lists(cohort: dict, rows: slice = slice(None)) -> List[List[List[str]]]:
    """``gen_comp_value`` output of the cohort's items: per item the level token lists."""
    tok, e, t = cohort["tok"][rows], cohort["entries"], cohort["tokens_per_entry"]
    out = []
    lex = cohort.get("lex")
    for row in tok:
        words = [f"t{int(v)}" for v in row] if lex is None else [lex[int(v)] for v in row]
        out.append([sorted(set(words[(e - 1 - lv) * t:]), key=str.casefold) for lv in range(e)])
    return out


def c5_category_lists(cohort: dict, rows: slice = slice(None)) -> List[List[str]]:
    return [[f"cat{b}" for b in range(64) if (int(m) >> b) & 1] for m in cohort["cat"][rows]]


# ------------------------------------------------------------------ "term": the reference's default configuration
_TERM_LETTERS = "enisratdhulcgmobwfkzvpjyxq"


def term_vocabulary(vocab: int = 5000, seed: int = 99) -> List[str]:
    """German-looking words: letters drawn with a skewed distribution, 3..12 letters."""
    rng = np.random.default_rng(seed)
    letters = np.array(list(_TERM_LETTERS))
    weight = 1.0 / np.arange(1, len(letters) + 1) ** 0.7
    weight /= weight.sum()
    return ["".join(rng.choice(letters, size=int(rng.integers(3, 13)), p=weight)) for _ in range(vocab)]


WORD_ALPHABET = " " + _TERM_LETTERS  # code units of the c5w level strings, in code order (blank = 0)


def word_vocabulary(vocab: int = 20_000, seed: int = 77, letters=(3, 7)) -> List[str]:
    """``vocab`` DISTINCT German-looking words of ``letters[0]..letters[1]`` letters (the skewed letter distribution
    of ``term_vocabulary``): eight of them joined with blanks stay within one 64-code-unit row."""
    rng = np.random.default_rng(seed)
    alpha = np.array(list(_TERM_LETTERS))
    weight = 1.0 / np.arange(1, len(alpha) + 1) ** 0.7
    weight /= weight.sum()
    seen, out = set(), []
    while len(out) < vocab:
        w = "".join(rng.choice(alpha, size=int(rng.integers(letters[0], letters[1] + 1)), p=weight))
        if w not in seen:
            seen.add(w)
            out.append(w)
    return out


def term_cohort(n: int, seed: int, vocab: int = 5000, entries=(3, 5), words=(2, 5), plant_from: list = None,
                plant_fraction: float = 0.01) -> List[List[List[str]]]:
    """Items shaped like the reference's ``Term`` column (config.yml:13 ``compare_column: Term``): a list of
    3..5 entries (sheet header, sub-headers, question, options -- types/questionnaire.py:59-68), each a few
    Zipf-distributed words.  Returns per item its entries as word lists; ``term_levels`` turns them into what
    ``gen_comp_value`` yields.  ``plant_from``: ``plant_fraction`` of the items are copies of random items of
    that cohort with one word replaced (near-duplicates, so that above-threshold pairs exist)."""
    rng = np.random.default_rng(seed)
    lex = term_vocabulary(vocab)
    zipf = 1.0 / np.arange(1, vocab + 1)
    zipf /= zipf.sum()
    n_entries = rng.integers(entries[0], entries[1] + 1, size=n)
    n_words = rng.integers(words[0], words[1] + 1, size=int(n_entries.sum()))
    picks = rng.choice(vocab, size=int(n_words.sum()), p=zipf)
    items, e_at, w_at = [], 0, 0
    for k in range(n):
        item = []
        for _ in range(int(n_entries[k])):
            c = int(n_words[e_at])
            item.append([lex[v] for v in picks[w_at: w_at + c]])
            e_at += 1
            w_at += c
        items.append(item)
    if plant_from is not None:
        for t in rng.choice(n, size=max(1, int(round(plant_fraction * n))), replace=False):
            src = [list(e) for e in plant_from[int(rng.integers(0, len(plant_from)))]]
            e = int(rng.integers(0, len(src)))
            src[e][int(rng.integers(0, len(src[e])))] = lex[int(rng.integers(0, vocab))]
            items[int(t)] = src
    return items


def term_levels(items: List[List[List[str]]]) -> List[List[List[str]]]:
    """``gen_comp_value`` of every item (types/comparable_data.py:283-285 with a whitespace tokenizer):
    level l = the sorted distinct words of the last l + 1 entries."""
    out = []
    for item in items:
        levels = []
        for lv in range(1, len(item) + 1):
            words = {w for entry in item[-lv:] for w in entry}
            levels.append(sorted(words, key=str.casefold))
        out.append(levels)
    return out


C5_ALPHABET = " t0123456789"  # code units of the C5 level strings, in code order


def c5_alphabet(cohort: dict) -> str:
    return C5_ALPHABET if cohort.get("lex") is None else WORD_ALPHABET


def c5_level_codes(cohort: dict, vocab: int = 20_000):
    """The fuzzy_match operands of a C5 cohort as dense code units, vectorised: what
    ``[[fuzzy_operand(level) for level in item] for item in c5_level_token_lists(cohort)]`` followed by
    ``Alphabet.encode`` yields, without building 4 n Python strings (configs[4]: 6 million).
    Returns ``codes`` uint8 [n * entries][64] (level l of item k is row k * entries + l; unused slots 0),
    ``lengths`` int32 [n * entries], ``first`` int32 [n], ``nlev`` int32 [n]; the alphabet is ``c5_alphabet(cohort)``
    (``C5_ALPHABET`` for ``t<id>`` words, ``WORD_ALPHABET`` for a cohort with a lexicon)."""
    tok, e, t = cohort["tok"], cohort["entries"], cohort["tokens_per_entry"]
    n = tok.shape[0]
    lex = cohort.get("lex")
    if lex is None:
        vocab = max(vocab, int(tok.max(initial=0)) + 1)
        names = [f"t{v}" for v in range(vocab)]
        alphabet = C5_ALPHABET
    else:
        names, vocab, alphabet = lex, len(lex), WORD_ALPHABET
    code_of = {ch: k for k, ch in enumerate(alphabet)}
    order = sorted(range(vocab), key=lambda v: names[v].casefold())  # (ties cannot occur: the words are distinct)
    rank = np.empty(vocab, dtype=np.int64)
    rank[order] = np.arange(vocab)
    wlen = np.array([len(s) for s in names], dtype=np.int64)
    wmax = int(wlen.max())
    wcode = np.zeros((vocab, wmax), dtype=np.uint8)
    for v, s in enumerate(names):
        wcode[v, : len(s)] = [code_of[ch] for ch in s]
    if e * t * (wmax + 1) - 1 > 64:
        raise NotImplementedError("level strings longer than one 64-code-unit row")
    codes = np.zeros((n * e, 64), dtype=np.uint8)
    lengths = np.zeros(n * e, dtype=np.int32)
    item = np.arange(n, dtype=np.int64)
    for lv in range(e):
        words = tok[:, (e - 1 - lv) * t:].astype(np.int64)  # the last lv + 1 entries
        srt = np.take_along_axis(words, np.argsort(rank[words], axis=1, kind="stable"), axis=1)
        keep = np.ones(srt.shape, dtype=bool)
        keep[:, 1:] = srt[:, 1:] != srt[:, :-1]  # sorted(set(...))
        span = np.where(keep, wlen[srt] + 1, 0)  # the word + the blank in front of the next word
        end = np.cumsum(span, axis=1)
        start = end - span
        rows = (item * e + lv)[:, None].repeat(srt.shape[1], axis=1)
        r, s, w = rows[keep], start[keep], srt[keep]
        for d in range(wmax):
            on = wlen[w] > d
            codes[r[on], s[on] + d] = wcode[w[on], d]
        # (the blank between two words has code 0: already there)
        lengths[item * e + lv] = (end[:, -1] - 1).astype(np.int32)
    first = (np.arange(n, dtype=np.int32) * e).astype(np.int32)
    return codes, lengths, first, np.full(n, e, dtype=np.int32)

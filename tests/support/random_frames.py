"""Random cohort frames, mappings and gen_comparable keyword sets for differential tests
(tools/fuzz_api.py on the GPU, tests/test_oracle_vs_reference.py against the reference itself)."""
import random

import pandas as pd

WORDS = [f"w{k}" for k in range(30)] + ["Dialyse", "nach", "Entlassung", "Fieber", "ja", "nein"]
COLUMNS = ["Identifier", "Term", "Tokens", "Variable", "Sheet", "Category"]


def frame(rng: random.Random, n: int, prefix: str, cat_kind: str, labels, empty_tokens: float = 0.02):
    rows = []
    for k in range(n):
        roll = rng.random()
        if roll < 0.05:
            toks = None
        elif roll < 0.05 + empty_tokens:
            toks = []
        else:
            toks = [" ".join(rng.choice(WORDS) for _ in range(rng.randint(1, 3))) for _ in range(rng.randint(1, 4))]
        cat = rng.sample(labels, rng.choice([0, 1, 1, 2])) if cat_kind == "list" else rng.choice(labels)
        rows.append({
            "Identifier": f"{prefix}{k if rng.random() > 0.05 else max(0, k - 1)}",
            "Term": [" ".join(rng.choice(WORDS) for _ in range(rng.randint(1, 4))) for _ in range(rng.randint(1, 3))],
            "Tokens": toks,
            "Variable": None if rng.random() < 0.05 else rng.choice(["gec_", "v_", ""]) + "".join(
                rng.choice("abcde") for _ in range(rng.randint(0, 7))),
            "Sheet": f"sheet{rng.randint(0, 2)}",
            "Category": cat,
        })
    return pd.DataFrame(rows, columns=COLUMNS)


def mapping(rng: random.Random, left, right, left_name: str, right_name: str):
    out = {}
    for k in range(rng.choice([0, 0, 1, 3])):
        entry = {}
        if len(left) and rng.random() < 0.9:
            entry[left_name] = [str(v) for v in rng.sample(list(left["Identifier"]), min(len(left), rng.randint(1, 2)))]
        if len(right) and rng.random() < 0.9:
            entry[right_name] = [str(v) for v in rng.sample(list(right["Identifier"]), min(len(right), rng.randint(1, 2)))]
        if rng.random() < 0.3:
            entry["other"] = ["x1"]
        out[f"uuid{k}"] = entry
    return out


def case(seed: int, score_funcs=("intersection_vs_union", "fuzzy_match"), sizes=(0, 1, 5, 5, 20, 20, 40, 40)):
    """(left, right, whitelist, blacklist, kwargs, category kinds) of round ``seed``."""
    rng = random.Random(seed)
    labels = [f"c{k}" for k in range(rng.choice([2, 5, 9]))]
    kinds = rng.choice([("list", "list"), ("list", "list"), ("scalar", "scalar"), ("scalar", "list"), ("list", "scalar")])
    empty_tokens = rng.choice([0.0, 0.0, 0.01, 0.05])  # an empty token list is a zero-level item: IndexError per pair
    left = frame(rng, rng.choice(sizes), "L", kinds[0], labels, empty_tokens)
    right = frame(rng, rng.choice(sizes), "R", kinds[1], labels, empty_tokens)
    wl, bl = mapping(rng, left, right, "hap", "suep"), mapping(rng, left, right, "hap", "suep")
    kw = dict(score_func=rng.choice(list(score_funcs)), compare_column=rng.choice(["Tokens", "Tokens", "Term", "Variable"]),
              left_name="hap", right_name="suep", filter_categories=rng.random() < 0.6,
              score_threshold=rng.choice([0.0, 0.05, 0.1, 0.3, 0.5, 0.7, 0.9]))
    return left, right, wl, bl, kw, kinds


def frames_differ(got, want, tol: float):
    """None when the two result frames agree (pair labels, columns, cells, scores within tol)."""
    if list(got.index) != list(want.index):
        return f"pair labels differ: {list(got.index)[:8]} vs {list(want.index)[:8]} ({len(got)} vs {len(want)})"
    if list(got.columns) != list(want.columns):
        return f"columns differ: {list(got.columns)} vs {list(want.columns)}"
    for col in want.columns:
        a, b = list(got[col]), list(want[col])
        if col == "MatchScore":
            if any(abs(x - y) > tol for x, y in zip(a, b)):
                return "scores differ"
        elif any(not (x == y or (x != x and y != y) or (x is None and y is None)) for x, y in zip(a, b)):
            return f"column {col} differs"
    return None

"""Oracle restatement of the pair grid (TEST INFRASTRUCTURE).

Restates, on plain pandas frames and plain dict mappings, what
``ComparableData.compare`` / ``gen_comparable`` / ``compare_terms`` do
(napkon_string_matching/types/comparable_data.py:69-299, :452-574) together with the
four ``Mapping`` lookups they use (types/mapping.py:66-72, :173-176, :200-203,
:281-289).  Pure-Python loops: only meant for grids of up to a few million pairs.
"""
from typing import Callable, Dict, Iterable, List, Optional, Sequence, Tuple

import pandas as pd

from . import score_functions

REMOVE_SYMBOLS = "!?,.()[]:;*"  # comparable_data.py:24
OUTPUT_COLUMNS = ["Identifier", "Argument", "Variable", "Sheet"]  # comparable.py:26-31
SCORE_COLUMN = "MatchScore"  # comparable.py:21

MappingDict = Dict[str, Dict[str, List[str]]]  # {uuid: {cohort: [identifier, ...]}}


# --------------------------------------------------------------------------- per item
def flatten_list(parts) -> List[str]:
    """comparable_data.py:567-574 -- one level of list flattening; a ``str`` input is
    iterated character by character."""
    out: List[str] = []
    for part in parts:
        if isinstance(part, list):
            out.extend(part)
        else:
            out.append(part)
    return out


def tokenize(
    parts,
    word_tokenize: Callable[[str], Iterable[str]] = str.split,
    stop_words: Iterable[str] = (),
) -> List[str]:
    """comparable_data.py:287-299.

    The reference calls nltk's punkt ``word_tokenize`` and drops German stop words;
    neither is available offline, so both are parameters.  With the defaults
    (``str.split``, no stop words) the result equals punkt's on inputs made of
    ``[A-Za-z0-9]+`` words that are not stop words.
    """
    words = word_tokenize(" ".join(flatten_list(parts)))
    stops = set(stop_words)
    kept = {w for w in words if w.casefold() not in stops and w not in REMOVE_SYMBOLS}
    return sorted(kept, key=str.casefold)


def gen_comp_value(items, **tok) -> List[List[str]]:
    """comparable_data.py:283-285 -- level l holds the tokens of the last l+1 entries."""
    return [tokenize(items[-k:], **tok) for k in range(1, len(items) + 1)]


# --------------------------------------------------------------------------- per pair
def compare_terms(left: Sequence, right: Sequence, score_func) -> float:
    """comparable_data.py:248-265 -- index starts at ONE, indices clamp to the last
    level, weights 1/2, 1/4, ...; a zero-level operand raises IndexError as soon as the
    other side has a level."""
    total = 0
    weight = 1
    last_l, last_r = len(left) - 1, len(right) - 1
    for step in range(1, max(len(left), len(right)) + 1):
        part = score_func(left[min(step, last_l)], right[min(step, last_r)])
        weight /= 2
        total += part * weight
    return total


def level_index_pairs(n_left: int, n_right: int) -> List[Tuple[int, int]]:
    """The (left level, right level) pairs ``compare_terms`` visits (for tests)."""
    return [
        (min(s, n_left - 1), min(s, n_right - 1)) for s in range(1, max(n_left, n_right) + 1)
    ]


def categories_predicate(first_left, first_right) -> Callable:
    """comparable_data.py:464-476 -- the predicate is picked from the TYPES IN ROW 0."""
    if isinstance(first_left, list):
        if isinstance(first_right, list):
            return lambda x, y: (not set(x).isdisjoint(set(y))) or (not x and not y)
        return lambda x, y: x in set(y)
    if isinstance(first_right, list):
        return lambda x, y: x in set(y)
    return lambda x, y: x == y


# --------------------------------------------------------------------------- mappings
def mapping_filter_by_group(mapping: MappingDict, group: str) -> Dict[str, List[str]]:
    """mapping.py:173-176 -- KeyError when ANY entry lacks ``group``."""
    return {key: entry[group] for key, entry in mapping.items() if entry[group]}


def existing_mapping_ids(identifiers: Sequence[str], group: str, mapping: MappingDict) -> List[str]:
    """comparable_data.py:452-461."""
    per_group = mapping_filter_by_group(mapping, group)
    found = {key for key, members in per_group.items() for ident in identifiers if ident in members}
    return list(found)


def whitelist_removals(
    left_ids: Sequence[str], right_ids: Sequence[str], left_name: str, right_name: str, whitelist: MappingDict
) -> Optional[Tuple[List[str], List[str]]]:
    """comparable_data.py:493-520 -- identifiers to drop on each side; None when the whole step is
    skipped on KeyError (:500-504)."""
    try:
        ids_l = existing_mapping_ids(left_ids, left_name, whitelist)
        ids_r = existing_mapping_ids(right_ids, right_name, whitelist)
    except KeyError:
        return None
    used = set(ids_l) & set(ids_r)
    kept = {key: entry for key, entry in whitelist.items() if key in used}  # mapping.py:200-203
    drop_l: List[str] = []
    drop_r: List[str] = []
    for entry in kept.values():
        drop_l += entry[left_name]
        drop_r += entry[right_name]
    return drop_l, drop_r


def blacklist_pairs(left_name: str, right_name: str, blacklist: MappingDict) -> List[Tuple[str, str]]:
    """comparable_data.py:555-564 + mapping.py:66-72,281-289 -- cartesian identifier pairs of
    every entry that has BOTH cohorts."""
    pairs: List[Tuple[str, str]] = []
    for entry in blacklist.values():
        if left_name in entry and right_name in entry:
            for a in entry[left_name]:
                for b in entry[right_name]:
                    pairs.append((a, b))
    return pairs


# --------------------------------------------------------------------------- the grid
def gen_comparable(
    left: pd.DataFrame,
    right: pd.DataFrame,
    whitelist: Optional[MappingDict],
    blacklist: Optional[MappingDict],
    score_func: str,
    compare_column: str,
    category_column: str = "Category",
    score_threshold: float = 0.1,
    left_name: str = None,
    right_name: str = None,
    filter_categories: bool = False,
    identifier_column_left: Optional[str] = None,
    identifier_column_right: Optional[str] = None,
    tokenizer: Optional[dict] = None,
    **_ignored,
) -> pd.DataFrame:
    """comparable_data.py:133-246, steps 1-12 of SURVEY.md section 3.2.

    Returns a frame indexed by the reference's pair label ``i*M' + j`` (positions after
    dropna / whitelist removal) with the surviving output columns and ``MatchScore``.
    """
    func = score_functions.get(score_func)  # :150
    tok = tokenizer or {}
    whitelist = whitelist or {}
    blacklist = blacklist or {}

    left = left.dropna(subset=[compare_column])  # :152
    right = right.dropna(subset=[compare_column])  # :153

    removals = whitelist_removals(  # :162-168
        list(left["Identifier"]), list(right["Identifier"]), left_name, right_name, whitelist
    )
    if removals is not None:  # None: the step was skipped as a whole (KeyError, :500-504)
        drop_l, drop_r = removals
        # :267-273 index the frame with a list of booleans; for a frame WITHOUT ROWS that list is empty,
        # pandas reads `frame[[]]` as "no columns", and the compare column is gone (KeyError below)
        left = left[[ident not in drop_l for ident in left["Identifier"]]]
        right = right[[ident not in drop_r for ident in right["Identifier"]]]

    lp, rp = left_name.title(), right_name.title()  # :186-187

    # :176-184 read the columns left-compare, right-compare, left-Term, right-Term in that order
    values = [list(left[compare_column]), list(right[compare_column])]  # KeyError on a column-less frame
    terms = [list(left["Term"]), list(right["Term"])]

    def side(frame: pd.DataFrame, prefix: str, compare_values, term_values):
        rows = []
        for (_, row), value, term in zip(frame.iterrows(), compare_values, term_values):
            rec = {prefix + col: row[col] for col in frame.columns}
            rec[prefix + "Compare"] = gen_comp_value(value, **tok)  # :176-177
            rec[prefix + "Argument"] = ":".join(flatten_list(term))  # :179-184
            rows.append(rec)
        return rows

    rows_l, rows_r = side(left, lp, values[0], terms[0]), side(right, rp, values[1], terms[1])
    n_right = len(rows_r)

    banned = blacklist_pairs(left_name, right_name, blacklist)  # :534
    id_l = lp + (identifier_column_left or "Identifier")  # :536-539
    id_r = rp + (identifier_column_right or "Identifier")

    grid = []  # (label, left row, right row), left-major as merge(how="cross") does (:191)
    for i, a in enumerate(rows_l):
        for j, b in enumerate(rows_r):
            if (a[id_l], b[id_r]) in banned:  # :542-552
                continue
            grid.append((i * n_right + j, a, b))

    # :191-206 -- the blacklist step also indexes with a list of booleans: an EMPTY cross join comes out
    # of it without columns (a cross join emptied by the blacklist keeps them)
    columnless = not rows_l or not rows_r
    if filter_categories:  # :209-218
        if not grid:
            raise IndexError("single positional indexer is out-of-bounds")  # df.iloc[0], :465
        _, a0, b0 = grid[0]
        pred = categories_predicate(a0[lp + category_column], b0[rp + category_column])
        grid = [g for g in grid if pred(g[1][lp + category_column], g[2][rp + category_column])]

    # :236-240 drops every column outside {Left,Right}x{Identifier,Argument,Variable,Sheet};
    # the survivors keep the FRAME's order (input order, then Argument which :179-184 appended).
    def kept(frame: pd.DataFrame, prefix: str) -> List[str]:
        return [prefix + c for c in list(frame.columns) + ["Compare", "Argument"] if c in OUTPUT_COLUMNS]

    if columnless:
        raise KeyError(lp + "Compare")  # :223-232 reads the compare columns of the column-less frame
    keep_cols = kept(left, lp) + kept(right, rp)
    out_index, out_rows = [], []
    for label, a, b in grid:
        score = compare_terms(a[lp + "Compare"], b[rp + "Compare"], func)  # :223-232
        if score >= score_threshold:  # :243
            rec = {c: (a[c] if c in a else b[c]) for c in keep_cols}
            rec[SCORE_COLUMN] = score
            out_index.append(label)
            out_rows.append(rec)
    return pd.DataFrame(out_rows, index=out_index, columns=keep_cols + [SCORE_COLUMN])


def compare(
    left: pd.DataFrame,
    right: pd.DataFrame,
    whitelist: Optional[MappingDict],
    blacklist: Optional[MappingDict],
    compare_column: str,
    score_threshold: float = 0.1,
    cache_threshold: Optional[float] = None,
    **kwargs,
) -> pd.DataFrame:
    """comparable_data.py:69-128 without the (address-dependent, never hitting) cache:
    score at ``cache_threshold or score_threshold`` (:102-108), keep ``>= score_threshold``
    (:123), order by score descending (:126).  Ties are put in the canonical order
    (label ascending); the reference's quicksort leaves them unspecified.
    """
    kwargs.pop("cached", None)
    kwargs.pop("cache_dir", None)
    first = cache_threshold if cache_threshold else score_threshold
    table = gen_comparable(
        left, right, whitelist, blacklist, compare_column=compare_column, score_threshold=first, **kwargs
    )
    table = table[table[SCORE_COLUMN] >= score_threshold]
    order = sorted(range(len(table)), key=lambda k: (-table[SCORE_COLUMN].iloc[k], table.index[k]))
    return table.iloc[order]


# --------------------------------------------------------------------------- raw grids
def raw_grid_hits(left_items: Sequence, right_items: Sequence, score_func: str, threshold: float):
    """RAW mode (what terminology/mesh.py:207-214 does 1xM): ``score_func(a, b)`` on one
    operand per item, hits ``>= threshold`` as (score, i, j) in canonical order."""
    func = score_functions.get(score_func)
    hits = []
    for i, a in enumerate(left_items):
        for j, b in enumerate(right_items):
            s = func(a, b)
            if s >= threshold:
                hits.append((s, i, j))
    hits.sort(key=lambda h: (-h[0], h[1], h[2]))
    return hits


def matcher_grid_hits(left_levels: Sequence, right_levels: Sequence, score_func: str, threshold: float):
    """MATCHER mode on pre-built level lists: ``compare_terms`` per pair."""
    func = score_functions.get(score_func)
    hits = []
    for i, a in enumerate(left_levels):
        for j, b in enumerate(right_levels):
            s = compare_terms(a, b, func)
            if s >= threshold:
                hits.append((s, i, j))
    hits.sort(key=lambda h: (-h[0], h[1], h[2]))
    return hits

"""Oracle restatement of the score_func plugin module (TEST INFRASTRUCTURE).

Follows napkon_string_matching/compare/score_functions.py:6-27 and, for the part
that lives in rapidfuzz 2.1.x, the published behaviour of ``fuzz.QRatio``:

    QRatio(s1, s2)  =  0                      if default_process(s1) or (s2) is empty
                       ratio(p1, p2)          otherwise
    ratio(p1, p2)   =  (1.0 - indel/(|p1|+|p2|)) * 100      (normalized Indel similarity)
    indel           =  |p1| + |p2| - 2 * LCS(p1, p2)

``fuzzy_match`` is *parity unpinned* (see oracle/__init__.py).
"""
from typing import List, Sequence, Union



def intersection_vs_union(left: Union[List[str], str], right: Union[List[str], str]) -> float:
    """score_functions.py:6-13 -- Jaccard of two token collections.

    A ``str`` operand is whitespace split first (:10-11).  Both empty raises
    ``ZeroDivisionError`` (:13); exactly one empty gives 0.0.
    """
    a = frozenset(left if isinstance(left, list) else left.split())
    b = frozenset(right if isinstance(right, list) else right.split())
    return len(a & b) / len(a | b)


def join_sorted(value: Sequence[str]) -> str:
    """score_functions.py:16-17 -- case-insensitively sorted, blank joined."""
    return " ".join(sorted(value, key=str.lower))


UNDERSCORE_POLICY = "blank"  # mirrors the product's switch (compare/score_functions.py); see default_process


def default_process(text: str, underscore: str = None) -> str:
    """rapidfuzz 2.x ``utils.default_process``: every non-alphanumeric code point becomes a blank, the
    result is stripped and lower-cased.

    Restated PER CODE POINT with ``str.isalnum`` -- on purpose not the regular expression the product
    uses, so that the parity test compares two implementations and not one pattern with itself.  The
    two implementations inside rapidfuzz 2.1 differ on "_" (C++: blank; pure-Python fallback
    ``re.sub(r"(?ui)\\W", " ", s)``: kept, because ``\\w`` is ``isalnum() or "_"``); ``underscore`` selects
    the reading ("blank" / "keep"), default = the compiled implementation's.  Not checkable offline:
    fuzzy_match stays *parity unpinned*.
    """
    policy = UNDERSCORE_POLICY if underscore is None else underscore
    if policy not in ("blank", "keep"):
        raise ValueError("underscore policy must be 'blank' or 'keep'")
    keep_underscore = policy == "keep"
    out = []
    for ch in text:
        out.append(ch if (ch.isalnum() or (keep_underscore and ch == "_")) else " ")
    return "".join(out).strip().lower()


def lcs_length(a: str, b: str) -> int:
    """Plain O(|a||b|) dynamic programme (deliberately NOT the bit-parallel form
    the HIP kernel uses)."""
    if not a or not b:
        return 0
    prev = [0] * (len(b) + 1)
    for ca in a:
        cur = [0]
        for j, cb in enumerate(b, 1):
            if ca == cb:
                cur.append(prev[j - 1] + 1)
            else:
                up, left_ = prev[j], cur[j - 1]
                cur.append(up if up >= left_ else left_)
        prev = cur
    return prev[-1]


def indel_ratio_from_lcs(len_a: int, len_b: int, lcs: int) -> float:
    """The float arithmetic of ``QRatio(...)/100`` once LCS is known.

    Kept as ONE function so the product's host code can be compared against the
    very same operation order: ((1 - dist/maximum) * 100) / 100 in IEEE double.
    """
    if len_a == 0 or len_b == 0:
        return 0 / 100  # QRatio returns integer 0 for an empty operand
    maximum = len_a + len_b
    dist = maximum - 2 * lcs
    norm_dist = dist / maximum
    norm_sim = 1.0 - norm_dist
    return (norm_sim * 100) / 100


def ratio(left: str, right: str) -> float:
    """``rapidfuzz.fuzz.ratio`` (no processor) in [0, 100]: the normalized Indel similarity of the strings
    as they are."""
    if not left or not right:
        return 0
    maximum = len(left) + len(right)
    dist = maximum - 2 * lcs_length(left, right)
    return (1.0 - dist / maximum) * 100


def q_ratio(left: str, right: str) -> float:
    """``rapidfuzz.fuzz.QRatio`` (2.x, processor=default_process) in [0, 100]."""
    a, b = default_process(left), default_process(right)
    if not a or not b:
        return 0
    maximum = len(a) + len(b)
    dist = maximum - 2 * lcs_length(a, b)
    return (1.0 - dist / maximum) * 100


def fuzzy_match(left: Union[str, List[str]], right: Union[str, List[str]]) -> float:
    """score_functions.py:20-27 -- lists are join_sorted first, then QRatio/100."""
    a = join_sorted(left) if isinstance(left, list) else left
    b = join_sorted(right) if isinstance(right, list) else right
    return q_ratio(a, b) / 100


SCORE_FUNCTIONS = {
    "intersection_vs_union": intersection_vs_union,
    "fuzzy_match": fuzzy_match,
}


def get(name: str):
    """comparable_data.py:150 resolves the plugin by attribute name; an unknown
    name is an ``AttributeError`` there, so it is one here."""
    try:
        return SCORE_FUNCTIONS[name]
    except KeyError:
        raise AttributeError(f"module 'score_functions' has no attribute '{name}'") from None

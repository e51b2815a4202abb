"""Oracle restatement of ``MeshProvider.get_matches`` (TEST INFRASTRUCTURE).

Follows napkon_string_matching/terminology/mesh.py:192-220: score every synonym with fuzzy_match
against ``" ".join(term)``, keep ``>= score_threshold``, order by score descending, keep the first row
per Id.  Ties are put in table order (the reference's quicksort leaves them unspecified).
"""
from typing import List, Sequence, Tuple

from .score_functions import fuzzy_match


def get_matches(ids: Sequence[str], terms: Sequence[str], term: Sequence[str], score_threshold: float = 0.1
                ) -> List[Tuple[str, str, float]]:
    joined = " ".join(term)
    scored = [(fuzzy_match(t, joined), k) for k, t in enumerate(terms)]
    kept = [(s, k) for s, k in scored if s >= score_threshold]
    kept.sort(key=lambda sk: (-sk[0], sk[1]))
    out, seen = [], set()
    for s, k in kept:
        if ids[k] in seen:
            continue
        seen.add(ids[k])
        out.append((ids[k], terms[k], s))
    return out

"""``MeshProvider.get_matches`` on the GPU (reference: napkon_string_matching/terminology/mesh.py:192-220).

The reference scores ONE item term against every MeSH synonym with ``np.vectorize(fuzzy_match)``
(1 x M, in a ``multiprocessing.Pool`` over the items, prepare/match_preparator.py:55-67).  That is the
same RAW ``fuzzy_match`` grid as the match loop, so all items go through one N x M launch here.
Database access (the Postgres MeSH dump, mesh.py:61-190) is out of scope: the synonym table is
handed in as a frame with the reference's column names ``Id`` and ``Term``.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import pandas as pd

from ..compare import score_functions

TERMINOLOGY_COLUMN_TERM = "Term"
TERMINOLOGY_COLUMN_ID = "Id"
TERMINOLOGY_COLUMN_SCORE = "Score"

Match = Tuple[str, str, float]


class MeshProvider:
    def __init__(self, config=None, synonyms: Optional[pd.DataFrame] = None, headings: Optional[pd.DataFrame] = None):
        self.config = config
        self._synonyms = synonyms
        self._headings = headings

    @property
    def initialized(self) -> bool:
        return self._synonyms is not None

    def initialize(self) -> None:
        if not self.initialized:
            raise RuntimeError("hand the synonym table to MeshProvider(synonyms=...): database access is out of scope")

    @property
    def synonyms(self) -> pd.DataFrame:
        return self._synonyms

    @property
    def headings(self) -> pd.DataFrame:
        return self._headings

    def get_matches(self, term: Sequence[str], score_threshold: float = 0.1) -> List[Match]:
        """(Id, Term, Score) of every synonym scoring ``>= score_threshold`` against ``" ".join(term)``,
        best first, one row per Id (mesh.py:207-220).  Equal scores keep the synonym table's order
        (the reference's quicksort leaves them unspecified)."""
        return self.get_matches_batch([term], score_threshold)[0]

    def get_matches_batch(self, terms: Sequence[Sequence[str]], score_threshold: float = 0.1) -> List[List[Match]]:
        syn = self.synonyms
        ids = list(syn[TERMINOLOGY_COLUMN_ID])
        syn_terms = list(syn[TERMINOLOGY_COLUMN_TERM])
        joined = [" ".join(term) for term in terms]  # mesh.py:207
        hits = score_functions.fuzzy_match.raw_grid(joined, syn_terms, score_threshold)
        out: List[List[Match]] = [[] for _ in terms]
        # hits arrive ordered by (score desc, item, synonym row): per item that is already
        # "score descending, table order among equals"
        seen = [set() for _ in terms]
        for score, i, j in zip(hits.score.tolist(), hits.i.tolist(), hits.j.tolist()):
            if ids[j] in seen[i]:
                continue  # drop_duplicates(subset=Id) keeps the best row of an Id
            seen[i].add(ids[j])
            out[i].append((ids[j], syn_terms[j], score))
        return out


class TerminologyProvider:
    """Combination of providers (reference: terminology/provider.py:11-55); ``None`` for no match."""

    def __init__(self, config=None, providers: Optional[Sequence[MeshProvider]] = None) -> None:
        self.config = config
        self.providers = list(providers or [])

    @property
    def initialized(self) -> bool:
        return all(p.initialized for p in self.providers)

    def initialize(self) -> None:
        for p in self.providers:
            p.initialize()

    def get_matches(self, term: Sequence[str], score_threshold: float = 0.1) -> Optional[List[Match]]:
        return self.get_matches_batch([term], score_threshold)[0]

    def get_matches_batch(self, terms, score_threshold: float = 0.1) -> List[Optional[List[Match]]]:
        merged: List[List[Match]] = [[] for _ in terms]
        for p in self.providers:
            for k, rows in enumerate(p.get_matches_batch(terms, score_threshold)):
                merged[k] += rows
        return [rows if rows else None for rows in merged]

"""``MatchPreparator.add_tokens`` (reference: napkon_string_matching/prepare/match_preparator.py:34-74):
every item's ``Term`` against every terminology synonym, in ONE grid launch instead of a process pool
of 1 x M ``np.vectorize`` calls."""
from __future__ import annotations

import logging

from ..terminology.mesh import TerminologyProvider

logger = logging.getLogger(__name__)


class MatchPreparator:
    def __init__(self, config=None, terminology_provider: TerminologyProvider = None):
        self.config = config
        self.terminology_provider = terminology_provider

    def add_tokens(self, cs, score_threshold: float = 0.1, verbose: bool = True, timeout=10) -> None:
        """Fills ``TokenIds`` / ``Tokens`` / ``TokenMatch`` of ``cs`` (match_preparator.py:69-73)."""
        del verbose, timeout  # one launch: nothing to show progress of, nothing to time out
        if self.terminology_provider is None or not self.terminology_provider.initialized:
            raise RuntimeError("'terms' and/or 'headings' not initialized")
        logger.info("add tokens...")
        results = self.terminology_provider.get_matches_batch(list(cs["Term"]), score_threshold)
        unpacked = [tuple(zip(*entry)) if entry else (None, None, None) for entry in results]
        cs["TokenIds"] = [ids if ids else None for ids, *_ in unpacked]
        cs["Tokens"] = [tokens if tokens else None for _, tokens, *_ in unpacked]
        cs["TokenMatch"] = results
        logger.info("...done")

"""``match(config)``: the entry point of the match loop (reference: napkon_string_matching/
matching.py:18-38).  ``config`` is the reference's YAML mapping (``matching:`` / ``steps:`` keys);
the loaded cohort tables are passed in because ingestion is out of scope."""
from __future__ import annotations

from typing import Dict

from .matcher import Matcher

CONFIG_FIELD_STEPS = "steps"


def create_matcher(config: Dict, use_cache: bool = True, **tables) -> Matcher:
    return Matcher(None, config, use_cache=use_cache, **tables)


def match(config: Dict, use_cache: bool = True, write: bool = True, **tables) -> Matcher:
    matcher = create_matcher(config, use_cache, **tables)
    for step in config[CONFIG_FIELD_STEPS]:
        if step == "variables":
            matcher.match_questionnaires_variables()
        elif step == "gecco":
            matcher.match_gecco_with_questionnaires()
        elif step == "questionnaires":
            matcher.match_questionnaires()
    matcher.print_analysis()
    if write:
        matcher.write_results()
    return matcher

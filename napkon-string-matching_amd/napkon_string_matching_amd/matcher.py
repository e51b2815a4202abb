"""``Matcher``: enumerates cohort pairs and runs the GPU match loop for each.

Mirrors napkon_string_matching/matcher.py:225-331 (``clear_results``, ``match_questionnaires``,
``match_questionnaires_variables``, ``match_gecco_with_questionnaires``, ``print_analysis``,
``write_results``): same pair enumeration (case-insensitively sorted names, no self / duplicate
pairs), same result keys, same keyword overrides.  The reference's constructor also ingests every
NAPKON spreadsheet (matcher.py:83-223); that is out of scope, so the already loaded tables are
handed in.
"""
from __future__ import annotations

import logging
from itertools import product
from pathlib import Path
from typing import Dict, Optional

from .types.comparable import ComparisonResults
from .types.comparable_data import ComparableData
from .types.mapping import Mapping

CONFIG_FIELD_MATCHING = "matching"
CONFIG_VARIABLE_THRESHOLD = "variable_score_threshold"
CONFIG_OUTPUT_DIR = "output_dir"
CONFIG_CACHE_DIR = "cache_dir"
RESULTS_FILE_PATTERN = "result_{score_threshold}_{compare_column}_{score_func}.xlsx"

logger = logging.getLogger(__name__)


class Matcher:
    def __init__(
        self,
        preparator,
        config: Dict,
        use_cache: bool = True,
        *,
        questionnaires: Optional[Dict[str, ComparableData]] = None,
        gecco: Optional[ComparableData] = None,
        mappings_whitelist: Optional[Mapping] = None,
        mappings_blacklist: Optional[Mapping] = None,
    ) -> None:
        self.preparator = preparator
        self.config = config
        self.use_cache = use_cache
        self.cache_dir = config.get(CONFIG_CACHE_DIR)
        self.questionnaires: Dict[str, ComparableData] = questionnaires if questionnaires is not None else {}
        self.gecco = gecco
        self.mappings_whitelist = mappings_whitelist if mappings_whitelist is not None else Mapping()
        self.mappings_blacklist = mappings_blacklist if mappings_blacklist is not None else Mapping()
        self.results: ComparisonResults = None
        self.clear_results()

    def clear_results(self) -> None:
        self.results = ComparisonResults()

    def match_gecco_with_questionnaires(self) -> None:
        for name, questionnaire in self.questionnaires.items():
            logger.info("compare gecco and %s", name)
            self.results[f"gecco vs {name}"] = self.gecco.compare(
                questionnaire,
                existing_mappings_whitelist=self.mappings_whitelist,
                existing_mappings_blacklist=self.mappings_blacklist,
                left_name="gecco",
                right_name=name,
                cache_dir=self.cache_dir,
                **self.config[CONFIG_FIELD_MATCHING],
            )

    def match_questionnaires(self, prefix: str = None, *args, **kwargs) -> None:
        from .types.comparable_data import ComparableData

        with ComparableData.item_memo():  # every cohort takes part in several grids
            self._match_questionnaires(prefix, **kwargs)

    def _match_questionnaires(self, prefix: str = None, **kwargs) -> None:
        done = set()
        for entry_a, entry_b in product(self.questionnaires.items(), self.questionnaires.items()):
            (name_first, data_first), (name_second, data_second) = sorted(
                [entry_a, entry_b], key=lambda entry: entry[0].lower()
            )
            if name_first == name_second:
                continue
            key = (name_first, name_second)
            if key in done:
                continue
            done.add(key)
            logger.info("compare %s %s and %s", prefix if prefix else "", name_first, name_second)
            self.results[f"{prefix if prefix else ''}{name_first} vs {name_second}"] = data_first.compare(
                data_second,
                existing_mappings_whitelist=self.mappings_whitelist,
                existing_mappings_blacklist=self.mappings_blacklist,
                left_name=name_first,
                right_name=name_second,
                cache_dir=self.cache_dir,
                **{**self.config[CONFIG_FIELD_MATCHING], **kwargs},
            )

    def match_questionnaires_variables(self) -> None:
        self.match_questionnaires(
            prefix="var_",
            compare_column="Variable",
            score_threshold=self.config[CONFIG_FIELD_MATCHING][CONFIG_VARIABLE_THRESHOLD],
        )

    # ---- result consumers (matcher.py:286-331)
    def _analyse(self) -> Dict[str, Dict[str, str]]:
        gecco_prefix = "gec_"
        out = {}
        for name, comp in self.results.items():
            if comp.empty:
                continue
            gecco_rows = comp[[gecco_prefix in entry for entry in comp.variable]]
            gecco_match_rows = comp[[gecco_prefix in entry for entry in comp.match_variable]]
            out[name] = {
                "matched": "{}/{}".format(comp.variable.nunique(), comp.match_variable.nunique()),
                "gecco": "{}/{}".format(gecco_rows.variable.nunique(), gecco_match_rows.match_variable.nunique()),
            }
        return out

    def print_analysis(self) -> None:
        for name, item in self._analyse().items():
            logger.info("%s\t%s", name, "\t".join(f"{k}: {v}" for k, v in item.items()))

    def write_results(self) -> None:
        """``result_{score_threshold}_{compare_column}_{score_func}.xlsx`` with one sheet per cohort
        pair (matcher.py:322-331); one CSV per cohort pair in a directory of that name when no Excel
        engine is installed."""
        matching = self.config[CONFIG_FIELD_MATCHING]
        name = RESULTS_FILE_PATTERN.format(**{**matching, "score_func": matching["score_func"].replace("_", "-")})
        out_dir = Path(self.config.get(CONFIG_OUTPUT_DIR) or ".")
        try:
            self.results.write_excel(out_dir / name)
        except ImportError:
            logger.warning("no Excel engine installed: writing CSV files instead")
            self.results.write_csv_dir(out_dir / name[: -len(".xlsx")])

"""Cohort item tables accepted by ``Matcher`` (reference: napkon_string_matching/types/
questionnaire.py:13-68, gecco_definition.py:39-42).  Only the column schema and ``add_terms`` are
kept; spreadsheet / JSON ingestion of the NAPKON files is out of scope (host keeps pandas frames)."""
from __future__ import annotations

from .comparable_data import ComparableData

QUESTIONNAIRE_COLUMNS = [
    "Term", "Tokens", "TokenIds", "TokenMatch", "Matches", "Identifier",  # ComparableColumns
    "Sheet", "File", "Header", "Question", "Options", "Variable", "Parameter", "Uid", "Category",
]


class Questionnaire(ComparableData):
    """hap / pop / suep style item table; ``Category`` holds a LIST of labels per item."""

    __column_mapping__ = {"Parameter": "Parameter"}

    def add_terms(self) -> None:
        """questionnaire.py:59-68 -- Term = [*header, question, parameter] without empty parts."""
        self._data["Term"] = [
            self.gen_term(*(header or ()), question, parameter)
            for header, question, parameter in zip(self._data["Header"], self._data["Question"], self._data["Parameter"])
        ]


class GeccoDefinition(ComparableData):
    """GECCO item table (gecco_definition.py:34-64); ``Category`` is a single label (a ``str``), which
    selects the ``x in set(y)`` branch of the category predicate when compared with a questionnaire."""

    __column_mapping__ = {}

    def map_for_comparable(self):
        """gecco_definition.py:41-44 -- the result's Variable column is the item's Identifier."""
        result = super().map_for_comparable().copy()
        result["Variable"] = result["Identifier"]
        return result

    def add_terms(self) -> None:
        """gecco_definition.py:57-64 -- Term = [category, parameter, choice] without empty parts."""
        self._data["Term"] = [
            self.gen_term(category, parameter, choice)
            for category, parameter, choice in zip(self._data["Category"], self._data["Parameter"], self._data["Choices"])
        ]

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
    """GECCO item table; ``Category`` is a single label (a ``str``), which selects the
    ``x in set(y)`` branch of the category predicate when compared with a questionnaire."""

    __column_mapping__ = {"Id": "Variable"}

"""Result containers of the match loop (reference: napkon_string_matching/types/comparable.py).

``Comparable`` = the above-threshold pairs of one cohort pair: ``left_name`` / ``right_name`` (the
title-cased column prefixes) and a DataFrame with ``{Left,Right}{Identifier,Argument,Variable,Sheet}``
and ``MatchScore``, indexed by the reference's pair label ``i*M + j``.
"""
from __future__ import annotations

import json
from pathlib import Path
from typing import Dict, Optional

import pandas as pd

IDENTIFIER, PARAMETER, VARIABLE, SHEET, MATCH_SCORE = "Identifier", "Parameter", "Variable", "Sheet", "MatchScore"
QUESTION_OUTPUT = "Argument"
COLUMN_NAMES = [IDENTIFIER, QUESTION_OUTPUT, VARIABLE, SHEET]  # comparable.py:26-31


class Comparable:
    def __init__(self, data=None, left_name: Optional[str] = None, right_name: Optional[str] = None) -> None:
        if left_name is not None and right_name is not None:
            frame = data
        elif isinstance(data, dict) and {"left_name", "right_name", "data"} <= set(data):
            left_name, right_name, frame = data["left_name"], data["right_name"], data["data"]
        else:
            raise AttributeError(
                "Either provide 'left_name' AND 'right_name' or a dictionary in 'data' providing the "
                "entries left_name, right_name AND data"
            )
        self.__dict__["left_name"] = left_name
        self.__dict__["right_name"] = right_name
        self.__dict__["data"] = frame if isinstance(frame, pd.DataFrame) else pd.DataFrame(frame)

    # -- the reference's attribute quirk (comparable.py:78-100): match_<col> -> LEFT, <col> -> RIGHT
    def _column_for(self, name: str) -> Optional[str]:
        if name == "match_score":
            return MATCH_SCORE
        parts = name.split("_")
        col = parts[-1].title()
        if col in COLUMN_NAMES:
            return (self.left_name if parts[0] == "match" else self.right_name) + col
        return None

    def __getattr__(self, name: str):
        col = self._column_for(name)
        if col is not None:
            return self.data[col]
        return getattr(self.data, name)

    def __setattr__(self, name: str, value) -> None:
        col = self._column_for(name)
        if col is not None:
            self.data[col] = value
        else:
            setattr(self.data, name, value)

    def __getitem__(self, item):
        got = self.data[item]
        if isinstance(got, pd.DataFrame):
            return Comparable(got, self.left_name, self.right_name)
        return got

    def __len__(self) -> int:
        return len(self.data)

    def __eq__(self, other) -> bool:
        return (
            isinstance(other, Comparable)
            and self.left_name == other.left_name
            and self.right_name == other.right_name
            and self.data.equals(other.data)
        )

    def __repr__(self) -> str:
        return repr(self.data)

    def __str__(self) -> str:
        return str(self.data)

    def dataframe(self) -> pd.DataFrame:
        return self.data

    # -- frame conveniences that keep the cohort names (comparable.py:119-145)
    def dropna(self, *args, **kwargs) -> "Comparable":
        return Comparable(self.data.dropna(*args, **kwargs), self.left_name, self.right_name)

    def drop(self, *args, **kwargs) -> "Comparable":
        return Comparable(self.data.drop(*args, **kwargs), self.left_name, self.right_name)

    def merge(self, *args, **kwargs) -> "Comparable":
        return Comparable(self.data.merge(*args, **kwargs), self.left_name, self.right_name)

    def drop_superfluous_columns(self, columns=None) -> None:
        """Keep only ``columns`` (default: the result columns of both sides + MatchScore), in place."""
        if columns is None:
            columns = [p + c for p in (self.left_name, self.right_name) for c in COLUMN_NAMES] + [MATCH_SCORE]
        extra = [c for c in self.data.columns if c not in set(columns)]
        self.__dict__["data"] = self.data.drop(columns=extra)

    def sort_by_score(self) -> None:
        """comparable.py:69-70, made deterministic: score descending, ties by pair label."""
        order = self.data.assign(_label=self.data.index).sort_values(
            by=[MATCH_SCORE, "_label"], ascending=[False, True], kind="mergesort"
        ).index
        self.__dict__["data"] = self.data.loc[order]

    def to_json(self, orient: Optional[str] = "records", **kwargs) -> str:
        payload = {"left_name": self.left_name, "right_name": self.right_name,
                   "data": self.data.to_dict(orient=orient)}
        return json.dumps(payload, **kwargs)

    def write_json(self, file_name) -> None:
        """Written to a temporary file and renamed into place: a concurrent reader (or a second writer of the
        same content, e.g. another rank of a sharded run) never sees a truncated file."""
        import os
        import tempfile

        target = Path(file_name)
        fd, tmp = tempfile.mkstemp(prefix=target.name + ".", suffix=".tmp", dir=str(target.parent))
        try:
            with os.fdopen(fd, "w", encoding="utf-8") as handle:
                handle.write(self.to_json(orient="records", indent=4))
            os.replace(tmp, target)
        except BaseException:
            if os.path.exists(tmp):
                os.unlink(tmp)
            raise

    @classmethod
    def read_json(cls, file_name) -> "Comparable":
        return cls(data=json.loads(Path(file_name).read_text(encoding="utf-8")))


class ComparisonResults:
    """``{result key: Comparable}`` (comparable.py:148-162); one spreadsheet sheet per key."""

    def __init__(self, comp_dict: Optional[Dict[str, Comparable]] = None) -> None:
        self.results: Dict[str, Comparable] = comp_dict if comp_dict else {}

    def __setitem__(self, key, value) -> None:
        self.results[key] = value

    def __getitem__(self, key) -> Comparable:
        return self.results[key]

    def __len__(self) -> int:
        return len(self.results)

    def items(self):
        return self.results.items()

    get_items = items

    def write_excel(self, file_name, *args, **kwargs) -> None:
        """One sheet per result key (reference: types/base/writable_excel.py:11-28, which still calls
        the ``writer.save()`` that pandas 2 removed).  Needs an Excel engine such as openpyxl."""
        path = Path(file_name)
        if path.parent and not path.parent.exists():
            path.parent.mkdir(parents=True)
        with pd.ExcelWriter(path) as writer:  # ImportError if no engine is installed
            for key, comp in self.results.items():
                comp.data.to_excel(writer, sheet_name=key[:31], index=False, *args, **kwargs)

    def write_csv_dir(self, directory) -> None:
        """One CSV per result key (the reference's xlsx writer needs openpyxl and the
        ``writer.save()`` API pandas 2 removed; spreadsheets are out of scope here)."""
        out = Path(directory)
        out.mkdir(parents=True, exist_ok=True)
        for key, comp in self.results.items():
            comp.data.to_csv(out / (key.replace(" ", "_") + ".csv"))

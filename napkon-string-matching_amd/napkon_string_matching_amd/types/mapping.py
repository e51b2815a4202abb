"""White / black lists ``{uuid: {cohort: [identifier, ...]}}``.

Only the lookups the match loop touches are provided (reference: napkon_string_matching/types/
mapping.py:66-72 ``get_group_combination``, :173-176 ``filter_by_group``, :200-203 ``get_filtered``,
:281-289 ``get_all_mapping_for_groups``); mapping curation (add / update / merge, Excel import)
is out of scope for this package.
"""
from __future__ import annotations

import json
from pathlib import Path
from typing import Dict, Iterable, List, Optional, Tuple


class MappingEntry:
    def __init__(self, data: Optional[Dict[str, List[str]]] = None) -> None:
        self._groups: Dict[str, List[str]] = data if data is not None else {}

    def __getitem__(self, group: str) -> List[str]:
        return self._groups[group]  # KeyError for an absent cohort, as upstream

    def get(self, group: str, default=None):
        return self._groups.get(group, default)

    def dict(self) -> Dict[str, List[str]]:
        return self._groups

    def get_group_names(self) -> List[str]:
        return list(self._groups)

    def get_group_combination(self, group_left: str, group_right: str) -> Optional[Tuple[List[str], List[str]]]:
        if group_left in self._groups and group_right in self._groups:
            return self._groups[group_left], self._groups[group_right]
        return None


class Mapping:
    def __init__(self, data: Optional[Dict[str, Dict[str, List[str]]]] = None) -> None:
        self._entries: Dict[str, MappingEntry] = {k: MappingEntry(v) for k, v in (data or {}).items()}

    @classmethod
    def read_json(cls, file_name) -> "Mapping":
        return cls(json.loads(Path(file_name).read_text(encoding="utf-8")))

    def __len__(self) -> int:
        return len(self._entries)

    def __iter__(self):
        return iter(self._entries.items())

    def items(self):
        return self._entries.items()

    def values(self):
        return self._entries.values()

    def dict(self) -> Dict[str, Dict[str, List[str]]]:
        return {k: e.dict() for k, e in self._entries.items()}

    def update(self, other: "Mapping") -> None:
        for key, entry in other.items():
            if key in self._entries:
                mine = self._entries[key].dict()
                for group, identifiers in entry.dict().items():
                    mine.setdefault(group, []).extend(identifiers)
            else:
                self._entries[key] = entry

    def filter_by_group(self, group: str) -> Dict[str, List[str]]:
        """Entries with a non-empty member list for ``group``; an entry WITHOUT the cohort key
        raises KeyError (which ``remove_existing_mappings`` upstream swallows, skipping the step)."""
        return {key: entry[group] for key, entry in self._entries.items() if entry[group]}

    def get_filtered(self, ids: Iterable[str]) -> "Mapping":
        wanted = set(ids)
        out = Mapping()
        out._entries = {k: e for k, e in self._entries.items() if k in wanted}
        return out

    def get_all_mapping_for_groups(self, group_left: str, group_right: str) -> List[Tuple[List[str], List[str]]]:
        combos = (e.get_group_combination(group_left, group_right) for e in self._entries.values())
        return [c for c in combos if c is not None]

"""score_func plugin surface (reference: napkon_string_matching/compare/)."""

"""Terminology lookups that reuse the all-pairs fuzzy grid."""

"""Token enrichment that reuses the all-pairs fuzzy grid."""

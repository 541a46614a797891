"""Top-level `layers` package: the reference's import paths (`layers.cloud_transform`,
`layers.multihead_ct`, ...), re-exported from the MI355X implementation so that
reference-style model_zoo files work unchanged (they are exec'd and import these names)."""

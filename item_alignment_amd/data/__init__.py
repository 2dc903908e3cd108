"""Tensor formats feeding the train step (the reference's collate outputs) and synthetic pair generators."""
from .synthetic import SyntheticCocaPairs, one_tower_text, two_tower_text  # noqa: F401

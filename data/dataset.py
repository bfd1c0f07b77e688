"""reference data/dataset.py surface (the one class the SML path imports) -> sml_amd.datasets."""
from sml_amd.datasets import offlineDataset_withsample  # noqa: F401

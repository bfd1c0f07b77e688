"""Drop-in `data` package (reference data/dataset2.py, data/dataset.py) -> sml_amd.datasets."""
